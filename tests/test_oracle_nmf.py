"""Pin the numpy restatement of the CD-NMF solver (oracle/nmf_cd.py) to the live
scikit-learn functions the reference calls, and to the known-answer vectors of
SURVEY.md section 8c.  CPU only."""
import numpy as np
import pytest

from cnmf_amd import synth
from oracle import nmf_cd, sklearn_ref


@pytest.fixture(scope="module")
def X():
    return synth.make_config("C1", dtype=np.float64, n_cells=400)


def test_known_answer_random_init():
    """_initialize_nmf(ones(4,3), 2, 'random', 59886188)."""
    z = np.random.RandomState(59886188).standard_normal(3)
    assert np.allclose(z, [0.29526446, 0.80487632, -0.3867717], atol=1e-8)
    W0, H0 = nmf_cd.random_init(np.ones((4, 3)), 2, 59886188)
    assert np.allclose(H0, [[0.208784, 0.569134, 0.273489], [0.696586, 0.211346, 1.095235]], atol=1e-6)
    assert np.allclose(W0[0], [0.525723, 1.104034], atol=1e-6)


def test_random_init_equals_sklearn(X):
    from sklearn.decomposition._nmf import _initialize_nmf
    for dt in (np.float64, np.float32):
        Xd = X.astype(dt)
        W_ref, H_ref = _initialize_nmf(Xd, 6, init="random", random_state=1234)
        W0, H0 = nmf_cd.random_init(Xd, 6, 1234)
        assert W0.dtype == dt and np.array_equal(W0, W_ref) and np.array_equal(H0, H_ref)


def test_known_answer_ledger():
    """SURVEY 8c ledger vectors for cNMF seed 14 (numpy legacy RNG is version-stable)."""
    led = sklearn_ref.ledger(range(5, 8), 15, 14)
    assert len(led) == 45
    assert [s for _, _, s in led[:3]] == [59886188, 1812018521, 1173234957]
    assert led[-1] == (7, 14, 1288885239)
    led = sklearn_ref.ledger(range(5, 14), 100, 14)
    assert len(led) == 900 and led[500] == (10, 0, 817790314) and led[-1] == (13, 99, 1393000)


def test_cd_sweep_equals_cython_kernel():
    from sklearn.decomposition._cdnmf_fast import _update_cdnmf_fast
    rs = np.random.RandomState(0)
    for dt in (np.float64, np.float32):
        W = np.abs(rs.standard_normal((50, 5))).astype(dt)
        W[rs.rand(50, 5) < 0.2] = 0
        Ht = np.abs(rs.standard_normal((30, 5))).astype(dt)
        HHt = Ht.T @ Ht
        XHt = np.abs(rs.standard_normal((50, 5))).astype(dt)
        W1, W2 = W.copy(), W.copy()
        v1 = _update_cdnmf_fast(W1, HHt, XHt, np.arange(5, dtype=np.intp))
        v2 = nmf_cd.cd_sweep(W2, HHt, XHt)
        tol = 1e-12 if dt == np.float64 else 1e-5
        assert np.allclose(W1, W2, rtol=tol, atol=tol)
        assert abs(v1 - v2) <= tol * abs(v1)


@pytest.mark.parametrize("k,seed", [(5, 3), (7, 59886188), (9, 11)])
def test_nmf_equals_sklearn(X, k, seed):
    H_ref, W_ref, n_ref = sklearn_ref.nmf(X, k, seed)
    W, H, n = nmf_cd.nmf(X, k, seed=seed)
    assert n == n_ref
    assert np.abs(H - H_ref).max() < 1e-10 and np.abs(W - W_ref).max() < 1e-9


def test_nmf_regularised_equals_sklearn(X):
    H_ref, W_ref, n_ref = sklearn_ref.nmf(X, 6, 42, alpha_W=0.002, alpha_H=0.001, l1_ratio=0.3)
    W, H, n = nmf_cd.nmf(X, 6, seed=42, alpha_W=0.002, alpha_H=0.001, l1_ratio=0.3)
    assert n == n_ref and np.abs(H - H_ref).max() < 1e-10


def test_nnls_equals_sklearn(X):
    _, H, _ = nmf_cd.nmf(X, 5, seed=1)
    Hn = H / H.sum(axis=1, keepdims=True)
    W_ref, n_ref = sklearn_ref.refit_usage(X, Hn)
    W, n = nmf_cd.nnls(X, Hn)
    assert n == n_ref and np.abs(W - W_ref).max() < 1e-9 * max(1.0, np.abs(W_ref).max())


def test_float32_path_tracks_float64(X):
    """The tolerance the GPU tests state is anchored here: sklearn's own float32 path vs
    float64 on the same seed stays within 1e-5 on normalised spectra for short restarts."""
    H64, _, n64 = sklearn_ref.nmf(X, 7, 5)
    H32, _, n32 = sklearn_ref.nmf(X.astype(np.float32), 7, 5)
    maxabs, relfro = nmf_cd.spectra_error(H64, H32)
    assert abs(n64 - n32) <= 2 and maxabs < 1e-4 and relfro < 1e-3
