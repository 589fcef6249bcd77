"""The host tail of the NNDSVD initialisation (cnmf_amd.engine._nndsvd_finish: small SVD, svd_flip, positive / negative
split on rows) against scikit-learn's _initialize_nmf, with the range finder taken from scikit-learn itself -- no device."""
import numpy as np
import pytest
from sklearn.decomposition._nmf import _initialize_nmf
from sklearn.utils.extmath import randomized_range_finder

from cnmf_amd.engine import _nndsvd_finish


@pytest.mark.parametrize("shape,k,seed", [((300, 60), 5, 3), ((60, 300), 4, 11), ((200, 200), 7, 42), ((90, 40), 6, 0)])
def test_finish_matches_sklearn_nndsvd(shape, k, seed):
    rs = np.random.RandomState(seed + 100)
    X = rs.gamma(0.6, 1.0, shape) * (rs.uniform(size=shape) < 0.7)
    W_ref, H_ref = _initialize_nmf(X, k, init="nndsvd", random_state=seed)
    transpose = X.shape[0] < X.shape[1]                       # sklearn utils/extmath.py:555-559
    M = X.T if transpose else X
    n_iter = 7 if k < 0.1 * min(X.shape) else 4
    Q = randomized_range_finder(M, size=k + 10, n_iter=n_iter, power_iteration_normalizer="auto", random_state=seed)
    W, H = _nndsvd_finish(k, Q, Q.T @ M, transpose)
    assert W.shape == W_ref.shape and H.shape == H_ref.shape and W.flags["C_CONTIGUOUS"]
    assert np.abs(W - W_ref).max() <= 1e-10 * np.abs(W_ref).max()
    assert np.abs(H - H_ref).max() <= 1e-10 * np.abs(H_ref).max()
    assert (W >= 0).all() and (H >= 0).all()
