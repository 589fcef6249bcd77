"""numpy restatement of scikit-learn's multiplicative-update NMF (TEST ORACLE).

The reference keeps ``solver='mu'`` whenever ``beta_loss != 'frobenius'``
(/root/reference/src/cnmf/cnmf.py:618-631); the arithmetic is sklearn 1.7.2:

* ``beta_divergence``  -> decomposition/_nmf.py:84-194   (dense X)
* ``mu_update_w``      -> decomposition/_nmf.py:526-631
* ``mu_update_h``      -> decomposition/_nmf.py:634-728
* ``fit_mu``           -> decomposition/_nmf.py:731-893  (error checked every 10 iterations)
* ``nmf_mu`` / ``nnls_mu`` -> non_negative_factorization(solver='mu'); update_H=False starts
  from W = full(avg) (decomposition/_nmf.py:1229-1231)

Dense X only, beta in {1 (Kullback-Leibler), 0 (Itakura-Saito)}; l1/l2 penalties supported.
Pinned against the live sklearn function in tests/test_oracle_mu.py.
"""
import numpy as np

from .nmf_cd import random_init, regularization

EPSILON = np.finfo(np.float32).eps


def beta_divergence(X, W, H, beta, square_root=False):
    WH = W @ H
    WH_data = WH.ravel()
    X_data = X.ravel()
    idx = X_data > EPSILON
    WH_data = WH_data[idx]
    X_data = X_data[idx]
    WH_data[WH_data < EPSILON] = EPSILON
    if beta == 1:
        sum_WH = np.dot(W.sum(axis=0), H.sum(axis=1))
        div = X_data / WH_data
        res = np.dot(X_data, np.log(div))
        res += sum_WH - X_data.sum()
    elif beta == 0:
        div = X_data / WH_data
        res = np.sum(div) - np.prod(X.shape) - np.sum(np.log(div))
    else:
        raise NotImplementedError
    if square_root:
        res = max(res, 0)
        return np.sqrt(2 * res)
    return res


def _ratio(X, W, H, beta):
    """numerator factor X * WH**(beta-2) and denominator factor WH**(beta-1) (None for beta=1)."""
    WH_safe = W @ H
    WH = WH_safe.copy()
    if beta - 1.0 < 0:
        WH[WH < EPSILON] = EPSILON
    WH_safe[WH_safe < EPSILON] = EPSILON
    if beta == 1:
        return X / WH_safe, None
    R = WH_safe ** -1
    R **= 2
    R *= X
    WH **= beta - 1
    return R, WH


def mu_update_w(X, W, H, beta, l1, l2, gamma):
    R, WHp = _ratio(X, W, H, beta)
    num = R @ H.T
    if beta == 1:
        den = np.sum(H, axis=1)[np.newaxis, :]
    else:
        den = WHp @ H.T
    if l1 > 0:
        den = den + l1
    if l2 > 0:
        den = den + l2 * W
    den = np.array(np.broadcast_to(den, W.shape))
    den[den == 0] = EPSILON
    num /= den
    if gamma != 1:
        num **= gamma
    W *= num
    return W


def mu_update_h(X, W, H, beta, l1, l2, gamma):
    R, WHp = _ratio(X, W, H, beta)
    num = W.T @ R
    if beta == 1:
        W_sum = np.sum(W, axis=0)
        W_sum[W_sum == 0] = 1.0
        den = W_sum[:, np.newaxis]
    else:
        den = W.T @ WHp
    if l1 > 0:
        den = den + l1
    if l2 > 0:
        den = den + l2 * H
    den = np.array(np.broadcast_to(den, H.shape))
    den[den == 0] = EPSILON
    num /= den
    if gamma != 1:
        num **= gamma
    H *= num
    return H


def fit_mu(X, W, H, beta, tol=1e-4, max_iter=200, l1W=0, l1H=0, l2W=0, l2H=0, update_H=True):
    gamma = 1.0 / (2.0 - beta) if beta < 1 else 1.0
    err0 = beta_divergence(X, W, H, beta, square_root=True)
    prev = err0
    n_iter = 0
    for n_iter in range(1, max_iter + 1):
        W = mu_update_w(X, W, H, beta, l1W, l2W, gamma)
        if beta < 1:
            W[W < np.finfo(np.float64).eps] = 0.0
        if update_H:
            H = mu_update_h(X, W, H, beta, l1H, l2H, gamma)
            if beta <= 1:
                H[H < np.finfo(np.float64).eps] = 0.0
        if tol > 0 and n_iter % 10 == 0:
            err = beta_divergence(X, W, H, beta, square_root=True)
            if (prev - err) / err0 < tol:
                break
            prev = err
    return W, H, n_iter


BETA = {"kullback-leibler": 1, "itakura-saito": 0, 1: 1, 0: 0}


def nmf_mu(X, n_components, seed=None, W0=None, H0=None, beta_loss="kullback-leibler", tol=1e-4,
           max_iter=1000, alpha_W=0.0, alpha_H=0.0, l1_ratio=0.0):
    X = np.asarray(X)
    if W0 is None:
        W, H = random_init(X, n_components, seed)
    else:
        W, H = np.array(W0, dtype=X.dtype), np.array(H0, dtype=X.dtype)
    l1W, l1H, l2W, l2H = regularization(X.shape[0], X.shape[1], alpha_W, alpha_H, l1_ratio)
    return fit_mu(X, W, H, BETA[beta_loss], tol, max_iter, l1W, l1H, l2W, l2H, True)


def nnls_mu(X, H, beta_loss="kullback-leibler", tol=1e-4, max_iter=1000, alpha_W=0.0, l1_ratio=0.0):
    X = np.asarray(X)
    H = np.asarray(H, dtype=X.dtype)
    avg = np.sqrt(X.mean() / H.shape[0])
    W = np.full((X.shape[0], H.shape[0]), avg, dtype=X.dtype)
    l1W, _, l2W, _ = regularization(X.shape[0], X.shape[1], alpha_W, 0.0, l1_ratio)
    W, _, n_iter = fit_mu(X, W, H, BETA[beta_loss], tol, max_iter, l1W, 0, l2W, 0, False)
    return W, n_iter
