import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


def _gpu_available():
    """A box with the AMD kernel driver node is a GPU box: there NOTHING is skipped -- a missing or broken
    libcnmf_hip.so must fail the tests loudly.  Only a box without /dev/kfd (e.g. the build container) skips."""
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    """`gpu` tests need a device: on a box without one they are skipped (missing hardware must not look like
    broken code).  On a GPU box nothing is skipped -- there the product fails loudly instead."""
    if not any("gpu" in it.keywords for it in items) or _gpu_available():
        return
    skip = pytest.mark.skip(reason="no /dev/kfd on this box (run with -m gpu on an MI355X)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def engine():
    """One engine (= one cnmf_ctx on cuda:0) shared by the GPU tests."""
    from cnmf_amd.engine import Engine
    eng = Engine(0)
    yield eng
    eng.close()
