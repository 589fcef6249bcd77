"""Pin the numpy restatement of sklearn's multiplicative-update NMF (oracle/nmf_mu.py),
the solver the reference keeps for beta_loss != 'frobenius' (cnmf.py:618-631).  CPU only."""
import numpy as np
import pytest

from cnmf_amd import synth
from oracle import nmf_mu, sklearn_ref


@pytest.fixture(scope="module")
def X():
    return synth.make_config("C1", dtype=np.float64, n_cells=300)[:, :200]


@pytest.mark.parametrize("beta_loss", ["kullback-leibler", "itakura-saito"])
def test_mu_equals_sklearn(X, beta_loss):
    Xp = X + (1e-3 if beta_loss == "itakura-saito" else 0.0)     # IS needs strictly positive data
    H_ref, W_ref, n_ref = sklearn_ref.nmf(Xp, 5, 7, beta_loss=beta_loss, solver="mu", max_iter=300)
    W, H, n = nmf_mu.nmf_mu(Xp, 5, seed=7, beta_loss=beta_loss, max_iter=300)
    assert n == n_ref
    assert np.abs(H - H_ref).max() <= 1e-9 * np.abs(H_ref).max()
    assert np.abs(W - W_ref).max() <= 1e-9 * np.abs(W_ref).max()


def test_mu_regularised_equals_sklearn(X):
    H_ref, W_ref, n_ref = sklearn_ref.nmf(X, 4, 3, beta_loss="kullback-leibler", solver="mu", max_iter=200,
                                          alpha_W=0.001, alpha_H=0.002, l1_ratio=0.5)
    W, H, n = nmf_mu.nmf_mu(X, 4, seed=3, max_iter=200, alpha_W=0.001, alpha_H=0.002, l1_ratio=0.5)
    assert n == n_ref and np.abs(H - H_ref).max() <= 1e-9 * np.abs(H_ref).max()


def test_mu_refit_equals_sklearn(X):
    _, H, _ = nmf_mu.nmf_mu(X, 4, seed=1, max_iter=100)
    Hn = H / H.sum(axis=1, keepdims=True)
    W_ref, n_ref = sklearn_ref.refit_usage(X, Hn, beta_loss="kullback-leibler", solver="mu", max_iter=200)
    W, n = nmf_mu.nnls_mu(X, Hn, max_iter=200)
    assert n == n_ref and np.abs(W - W_ref).max() <= 1e-9 * np.abs(W_ref).max()


def test_kl_on_scipy_sparse_input_equals_the_dense_restatement():
    """scikit-learn's Kullback-Leibler updates for scipy.sparse input touch only the stored entries
    (`_special_sparse_dot`, _nmf.py:84-194, 526-728) -- the path cNMF takes when the normalised counts are stored sparse
    and the one `kernels_mu_sparse.hip.h` restates on the device.  It is the same mathematics as the dense formulas (the
    quotient vanishes where X does), so the dense numpy restatement is a valid checker for the device's non-zero path:
    pinned here on a matrix with ~16 % non-zeros, restarts and a refit."""
    import scipy.sparse as sp
    C, _ = synth.topic_counts(400, 300, 6, 4.0, 0.4, 5)
    Xs = synth.normalise_like_prepare(C, dtype=np.float64)
    assert (Xs != 0).mean() < 0.2
    csr = sp.csr_matrix(Xs)
    for k, seed in ((5, 7), (9, 3)):
        H_ref, W_ref, n_ref = sklearn_ref.nmf(csr, k, seed, beta_loss="kullback-leibler", solver="mu", max_iter=200)
        W, H, n = nmf_mu.nmf_mu(Xs, k, seed=seed, max_iter=200)
        assert n == n_ref
        assert np.abs(H - H_ref).max() <= 1e-8 * np.abs(H_ref).max()
        assert np.abs(W - W_ref).max() <= 1e-8 * np.abs(W_ref).max()
    Hn = H / H.sum(axis=1, keepdims=True)
    W_ref, n_ref = sklearn_ref.refit_usage(csr, Hn, beta_loss="kullback-leibler", solver="mu", max_iter=200)
    W, n = nmf_mu.nnls_mu(Xs, Hn, max_iter=200)
    assert n == n_ref and np.abs(W - W_ref).max() <= 1e-8 * np.abs(W_ref).max()


def test_csr_walking_restatement_equals_sklearn_on_csr():
    """oracle/nmf_mu_csr.py (round 5: scikit-learn's Kullback-Leibler updates on the STORED entries, component by component
    over flat index arrays -- what generated nothing by itself but checks the full-size golden `ref_c4_kl.npz`, which
    scikit-learn itself produced) against the live scikit-learn function on CSR input: restarts with and without penalties,
    iteration counts, the divergence."""
    import scipy.sparse as sp
    from oracle import nmf_mu_csr
    C, _ = synth.topic_counts(900, 500, 6, 4.2, 0.4, 8)
    Xd = synth.normalise_like_prepare(C, dtype=np.float64)
    csr = sp.csr_matrix(Xd)
    for k, seed, kw in ((5, 7, {}), (12, 3, {}), (6, 4, dict(alpha_W=0.001, alpha_H=0.002, l1_ratio=0.5))):
        H_ref, W_ref, n_ref = sklearn_ref.nmf(csr, k, seed, beta_loss="kullback-leibler", solver="mu", max_iter=200, **kw)
        W, H, n = nmf_mu_csr.nmf_kl_csr(csr, k, seed, max_iter=200, **kw)
        assert n == n_ref
        assert np.abs(H - H_ref).max() <= 1e-9 * np.abs(H_ref).max() and np.abs(W - W_ref).max() <= 1e-9 * np.abs(W_ref).max()
        ii, jj = csr.nonzero()
        assert abs(nmf_mu_csr.kl_divergence(csr, W, H, ii, jj) - nmf_mu.beta_divergence(Xd, W, H, 1, square_root=True)) <= 1e-9
