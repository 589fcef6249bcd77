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
