"""numpy restatement of scikit-learn's coordinate-descent NMF (TEST ORACLE).

Follows, in behaviour, sklearn 1.7.2 (installed at
/usr/local/lib/python3.10/dist-packages/sklearn):

* ``random_init``            -> decomposition/_nmf.py:302-314 (``_initialize_nmf``,
                               init='random'; H is drawn BEFORE W)
* ``cd_sweep``               -> decomposition/_cdnmf_fast.pyx:8-38
* ``update_coordinate_descent`` -> decomposition/_nmf.py:376-403
* ``fit_coordinate_descent`` -> decomposition/_nmf.py:406-523
* ``regularization``         -> decomposition/_nmf.py:1254-1265
* ``nmf`` / ``nnls``         -> ``non_negative_factorization`` :905-1131 with
                               update_H True / False (``_check_w_h`` :1194-1252)

which is what the reference reaches through ``cNMF._nmf``
(/root/reference/src/cnmf/cnmf.py:661-674) and ``refit_usage`` (:776-802).

The sweep is vectorised over rows (rows are independent inside one component
``t``); the order over components and the use of already-updated columns
``r < t`` are exactly the Cython kernel's.  The only arithmetic difference is
summation order inside ``grad`` (BLAS matvec vs. the scalar loop) and inside
``violation`` (pairwise ``np.sum`` vs. running scalar sum).

Pinned by tests/test_oracle_nmf.py against the live sklearn function.
"""
import numpy as np


def regularization(n_samples, n_features, alpha_W=0.0, alpha_H=0.0, l1_ratio=0.0):
    """sklearn _nmf.py:1254-1265 (``alpha_H='same'`` is never used by cNMF)."""
    l1_reg_W = n_features * alpha_W * l1_ratio
    l1_reg_H = n_samples * alpha_H * l1_ratio
    l2_reg_W = n_features * alpha_W * (1.0 - l1_ratio)
    l2_reg_H = n_samples * alpha_H * (1.0 - l1_ratio)
    return l1_reg_W, l1_reg_H, l2_reg_W, l2_reg_H


def random_init(X, n_components, seed):
    """sklearn _nmf.py:302-314: avg*|randn|, H first then W, cast to X.dtype."""
    avg = np.sqrt(X.mean() / n_components)
    rng = np.random.RandomState(seed)
    n_samples, n_features = X.shape
    H = avg * rng.standard_normal(size=(n_components, n_features)).astype(X.dtype, copy=False)
    W = avg * rng.standard_normal(size=(n_samples, n_components)).astype(X.dtype, copy=False)
    np.abs(H, out=H)
    np.abs(W, out=W)
    return W, H


def cd_sweep(W, HHt, XHt):
    """One in-place cyclic CD sweep over the columns of W (_cdnmf_fast.pyx:18-36).

    Returns the projected-gradient violation (sum |pg|)."""
    k = W.shape[1]
    violation = 0.0
    for t in range(k):
        # grad = -XHt[i,t] + sum_r HHt[t,r] * W[i,r]   (uses updated W[:, r<t])
        grad = W @ HHt[t, :] - XHt[:, t]
        wt = W[:, t]
        pg = np.where(wt == 0, np.minimum(0.0, grad), grad)
        violation += float(np.abs(pg).sum(dtype=np.float64))
        hess = HHt[t, t]
        if hess != 0:
            W[:, t] = np.maximum(wt - grad / hess, 0.0)
    return violation


def update_coordinate_descent(X, W, Ht, l1_reg, l2_reg):
    """sklearn _nmf.py:376-403 with shuffle=False."""
    k = Ht.shape[1]
    HHt = Ht.T @ Ht
    XHt = X @ Ht
    if l2_reg != 0.0:
        HHt.flat[:: k + 1] += l2_reg
    if l1_reg != 0.0:
        XHt -= l1_reg
    return cd_sweep(W, HHt, XHt)


def fit_coordinate_descent(X, W, H, tol=1e-4, max_iter=200, l1_reg_W=0, l1_reg_H=0,
                           l2_reg_W=0, l2_reg_H=0, update_H=True, trace=None):
    """sklearn _nmf.py:406-523.  W is updated in place; returns (W, H, n_iter)."""
    Ht = np.ascontiguousarray(H.T)
    Xt = X.T
    n_iter = 0
    violation_init = None
    for n_iter in range(1, max_iter + 1):
        violation = 0.0
        violation += update_coordinate_descent(X, W, Ht, l1_reg_W, l2_reg_W)
        if update_H:
            violation += update_coordinate_descent(Xt, Ht, W, l1_reg_H, l2_reg_H)
        if n_iter == 1:
            violation_init = violation
        if trace is not None:
            trace.append(violation)
        if violation_init == 0:
            break
        if violation / violation_init <= tol:
            break
    return W, Ht.T, n_iter


def nmf(X, n_components, seed=None, W0=None, H0=None, tol=1e-4, max_iter=1000,
        alpha_W=0.0, alpha_H=0.0, l1_ratio=0.0, trace=None):
    """CD NMF with both factors free: ``non_negative_factorization(X, n_components=k,
    init='random', solver='cd', beta_loss='frobenius', ...)``.

    Returns (W, H, n_iter) -- the same order as sklearn."""
    X = np.asarray(X)
    if W0 is None or H0 is None:
        W, H = random_init(X, n_components, seed)
    else:
        W, H = np.array(W0, dtype=X.dtype), np.array(H0, dtype=X.dtype)
    l1W, l1H, l2W, l2H = regularization(X.shape[0], X.shape[1], alpha_W, alpha_H, l1_ratio)
    return fit_coordinate_descent(X, W, H, tol, max_iter, l1W, l1H, l2W, l2H, True, trace)


def nnls(X, H, tol=1e-4, max_iter=1000, alpha_W=0.0, l1_ratio=0.0):
    """``update_H=False`` refit (sklearn _nmf.py:1210-1233: W0 = zeros for 'cd').

    Returns (W, n_iter)."""
    X = np.asarray(X)
    H = np.asarray(H, dtype=X.dtype)
    W = np.zeros((X.shape[0], H.shape[0]), dtype=X.dtype)
    l1W, _, l2W, _ = regularization(X.shape[0], X.shape[1], alpha_W, 0.0, l1_ratio)
    W, _, n_iter = fit_coordinate_descent(X, W, H, tol, max_iter, l1W, 0, l2W, 0, False)
    return W, n_iter


# ---------------------------------------------------------------- comparison
def match_components(H_ref, H_test):
    """Greedy best-cosine matching of rows of H_test to rows of H_ref.

    Returns (perm, cos) with H_test[perm[i]] matched to H_ref[i]."""
    a = H_ref / np.maximum(np.linalg.norm(H_ref, axis=1, keepdims=True), 1e-300)
    b = H_test / np.maximum(np.linalg.norm(H_test, axis=1, keepdims=True), 1e-300)
    C = a @ b.T
    k = C.shape[0]
    perm = -np.ones(k, dtype=int)
    cos = np.zeros(k)
    Cw = C.copy()
    for _ in range(k):
        i, j = np.unravel_index(np.argmax(Cw), Cw.shape)
        perm[i] = j
        cos[i] = C[i, j]
        Cw[i, :] = -np.inf
        Cw[:, j] = -np.inf
    return perm, cos


def spectra_error(H_ref, H_test):
    """Parity measure of SURVEY 8c: L2-normalise rows, match by cosine, report
    (max-abs difference, relative Frobenius difference)."""
    H_ref = np.asarray(H_ref, dtype=np.float64)
    H_test = np.asarray(H_test, dtype=np.float64)
    perm, _ = match_components(H_ref, H_test)
    a = H_ref / np.maximum(np.linalg.norm(H_ref, axis=1, keepdims=True), 1e-300)
    b = H_test[perm] / np.maximum(np.linalg.norm(H_test[perm], axis=1, keepdims=True), 1e-300)
    return float(np.abs(a - b).max()), float(np.linalg.norm(a - b) / np.linalg.norm(a))
