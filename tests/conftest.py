import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def engine():
    """One engine (= one cnmf_ctx on cuda:0) shared by the GPU tests."""
    from cnmf_amd.engine import Engine
    eng = Engine(0)
    yield eng
    eng.close()
