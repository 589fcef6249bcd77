"""Pins oracle/mt19937.c (the C restatement of numpy's legacy MT19937 + polar gauss, built by `make -C oracle` /
`__graft_entry__.build()`) against numpy itself and against the SURVEY known-answer vectors.  The device RNG
(cnmf_amd/csrc/kernels_rng.hip.h) is pinned against numpy in tests/test_gpu_nmf.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ORACLE_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")


@pytest.fixture(scope="module")
def lib():
    subprocess.run(["make", "-C", ORACLE_DIR], check=True, capture_output=True)
    lib = C.CDLL(os.path.join(ORACLE_DIR, "libcnmf_oracle.so"))
    lib.oracle_standard_normal.argtypes = [C.c_uint32, C.c_int64, C.POINTER(C.c_double)]
    lib.oracle_random_init.argtypes = [C.c_uint32, C.c_double, C.c_int, C.c_int64, C.c_int64,
                                       C.POINTER(C.c_float), C.POINTER(C.c_float)]
    return lib


def test_standard_normal_equals_numpy(lib):
    for seed, n in [(59886188, 3), (1, 1001), (2 ** 31 - 2, 5000), (1812018521, 1249)]:
        out = np.empty(n)
        lib.oracle_standard_normal(seed, n, out.ctypes.data_as(C.POINTER(C.c_double)))
        ref = np.random.RandomState(seed).standard_normal(n)
        # same algorithm, libm's log/sqrt: bit-equal almost everywhere, never more than an ulp or two
        assert np.max(np.abs(out - ref) / np.maximum(np.abs(ref), 1e-300)) < 1e-15
        assert np.mean(out == ref) > 0.9
    out = np.empty(3)
    lib.oracle_standard_normal(59886188, 3, out.ctypes.data_as(C.POINTER(C.c_double)))
    assert np.allclose(out, [0.29526446, 0.80487632, -0.3867717], atol=1e-8)


def test_random_init_equals_sklearn_order(lib):
    """H (k x G) is drawn BEFORE W (N x k) from one RandomState(seed) (sklearn _nmf.py:303-314)."""
    from oracle import nmf_cd
    X = np.ones((4, 3))
    W_ref, H_ref = nmf_cd.random_init(X, 2, 59886188)
    W0, H0 = np.empty((4, 2), np.float32), np.empty((2, 3), np.float32)
    lib.oracle_random_init(59886188, float(np.sqrt(X.mean() / 2)), 2, 4, 3, W0.ctypes.data_as(C.POINTER(C.c_float)),
                           H0.ctypes.data_as(C.POINTER(C.c_float)))
    assert np.allclose(H0, H_ref, rtol=1e-6) and np.allclose(W0, W_ref, rtol=1e-6)
    assert np.allclose(H0, [[0.208784, 0.569134, 0.273489], [0.696586, 0.211346, 1.095235]], atol=1e-6)
