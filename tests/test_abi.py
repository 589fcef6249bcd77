"""The C-ABI library loads and exports every symbol include/cnmf_hip.h declares; argument
checking that needs no GPU.  CPU only (no compute calls)."""
import ctypes as C
import os
import re

import pytest

from cnmf_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    _lib.build()
    return _lib.load()


def header_symbols(name="cnmf_hip.h"):
    src = open(os.path.join(ROOT, "include", name)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cnmf_[a-z_0-9]+)\s*\(", src)))


def test_header_and_loader_agree():
    assert header_symbols() == sorted(_lib.SYMBOLS)
    # the diagnostic entry points live in their own header, behind a build flag: none of them in the drop-in boundary
    assert not [s for s in header_symbols() if s.startswith("cnmf_debug_")]
    assert header_symbols("cnmf_hip_debug.h") == sorted(_lib.DEBUG_SYMBOLS)
    assert "#ifdef CNMF_DEBUG_ABI" in open(os.path.join(ROOT, "cnmf_amd", "csrc", "debug_host.hip.h")).read()


def test_every_declared_symbol_is_exported(lib):
    missing = [s for s in header_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    # the in-tree build carries the test hooks (tests/ call them); a product build (CNMF_PRODUCT_BUILD=1) has none
    assert lib.has_debug_abi and not [s for s in _lib.DEBUG_SYMBOLS if not hasattr(lib, s)]


def test_version_and_error_strings(lib):
    assert b"gfx950" in lib.cnmf_version()
    assert isinstance(lib.cnmf_last_error(None), bytes)


def test_struct_layouts_match_header():
    """ctypes mirrors of the parameter structs (sizes as laid out by the C compiler)."""
    import subprocess
    import tempfile
    prog = r'''
    #include <stdio.h>
    #include "cnmf_hip.h"
    int main(void){ printf("%zu %zu %zu\n", sizeof(cnmf_cd_params), sizeof(cnmf_batch_stats), sizeof(cnmf_consensus_params)); return 0; }
    '''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(prog)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")], check=True)
        out = subprocess.run([os.path.join(d, "t")], capture_output=True, text=True, check=True).stdout.split()
    assert [int(x) for x in out] == [C.sizeof(_lib.CdParams), C.sizeof(_lib.BatchStats), C.sizeof(_lib.ConsensusParams)]


def test_no_device_fails_loudly(lib):
    """On a box without a GPU the product path must raise, never fall back."""
    if lib.cnmf_device_count() > 0:
        pytest.skip("a GPU is visible")
    assert lib.cnmf_create(0) is None
    assert b"no HIP device" in lib.cnmf_last_error(None)
    from cnmf_amd.engine import Engine
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Engine(0)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under cnmf_amd/ may import it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "cnmf_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f
                assert not re.search(r"^\s*(from|import)\s+sklearn", src, flags=re.M), f
