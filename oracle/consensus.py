"""numpy restatement of the consensus core (TEST ORACLE).

Follows, in behaviour, the reference ``cNMF.consensus`` core
(/root/reference/src/cnmf/cnmf.py:871-936) and the third-party functions it calls
(scikit-learn 1.7.2, pandas 2.3):

* ``l2_normalise``        -> cnmf.py:882
* ``euclidean_distances`` -> sklearn/metrics/pairwise.py:391-438 (float64 branch)
* ``local_density``       -> cnmf.py:879,893-898 (argpartition + sum / n)
* ``kmeans``              -> sklearn/cluster/_kmeans.py:1427-1555 (fit), :174-272
                             (k-means++), :624-752 (single Lloyd run), :279-288 (tol);
                             sklearn/cluster/_k_means_lloyd.pyx:26-219,
                             _k_means_common.pyx:167-211 (empty clusters), :270-330
* ``groupby_median``      -> pandas DataFrame.groupby(labels).median()  (cnmf.py:913)
* ``silhouette_score``    -> sklearn/metrics/cluster/_unsupervised.py:141-201 (+samples)
* ``consensus_core``      -> cnmf.py:871-936 end to end (NNLS refit via oracle.nmf_cd)

Pinned against the live sklearn / pandas functions in tests/test_oracle_consensus.py and
against the unmodified reference (run through oracle/scanpy_shim.py) by the fixtures in
tests/golden/ (tools/make_golden.py).
"""
import numpy as np

from . import nmf_cd


def l2_normalise(S):
    S = np.asarray(S, dtype=np.float64)
    return (S.T / np.sqrt((S ** 2).sum(axis=1))).T


def euclidean_distances(X, squared=False):
    X = np.asarray(X, dtype=np.float64)
    XX = (X * X).sum(axis=1)[:, None]
    D = -2.0 * (X @ X.T)
    D += XX
    D += XX.T
    np.maximum(D, 0, out=D)
    np.fill_diagonal(D, 0.0)
    return D if squared else np.sqrt(D)


def local_density(D, n_neighbors):
    """Mean distance to the n nearest neighbours: sum of the n+1 smallest entries of every
    row (self = 0 included) divided by n."""
    part = np.partition(D, n_neighbors + 1, axis=1)[:, :n_neighbors + 1]
    return part.sum(axis=1) / n_neighbors


# ------------------------------------------------------------------ KMeans
def _sq_dists(C, X, x_sq):
    """sklearn _euclidean_distances(C, X, Y_norm_squared=x_sq, squared=True)."""
    D = -2.0 * (C @ X.T)
    D += (C * C).sum(axis=1)[:, None]
    D += x_sq[None, :]
    np.maximum(D, 0, out=D)
    return D


def kmeans_plusplus(X, k, x_sq, rng):
    n = X.shape[0]
    n_local_trials = 2 + int(np.log(k))
    centers = np.empty((k, X.shape[1]), dtype=X.dtype)
    indices = np.full(k, -1, dtype=int)
    cid = rng.choice(n, p=np.ones(n) / n)
    centers[0] = X[cid]
    indices[0] = cid
    closest = _sq_dists(centers[0:1], X, x_sq)
    pot = closest @ np.ones(n)
    for c in range(1, k):
        rand_vals = rng.uniform(size=n_local_trials) * pot
        cand = np.searchsorted(np.cumsum(closest.ravel(), dtype=np.float64), rand_vals)
        np.clip(cand, None, n - 1, out=cand)
        d = _sq_dists(X[cand], X, x_sq)
        np.minimum(closest, d, out=d)
        cpot = d @ np.ones((n, 1))
        best = int(np.argmin(cpot))
        pot = cpot[best]
        closest = d[best:best + 1]
        centers[c] = X[cand[best]]
        indices[c] = cand[best]
    return centers, indices


def lloyd_iter(X, centers, update=True):
    """One E(+M) step (sklearn _k_means_lloyd.pyx): labels by first-min of
    ||c||^2 - 2 x.c; new centres = means; empty clusters relocated to the farthest points."""
    k = centers.shape[0]
    csq = (centers * centers).sum(axis=1)
    pd = csq[None, :] - 2.0 * (X @ centers.T)
    labels = np.argmin(pd, axis=1).astype(np.int32)
    if not update:
        return labels, None, None
    w = np.bincount(labels, minlength=k).astype(X.dtype)
    new = np.zeros_like(centers)
    np.add.at(new, labels, X)
    empty = np.where(w == 0)[0]
    if empty.size:
        dist = ((X - centers[labels]) ** 2).sum(axis=1)
        if dist.max() > 0:
            far = np.argpartition(dist, -empty.size)[:-empty.size - 1:-1]
            for idx, e in enumerate(empty):
                f = far[idx]
                old = labels[f]
                new[old] -= X[f]
                new[e] = X[f]
                w[e] = 1
                w[old] -= 1
    amax = int(np.argmax(w))
    for j in range(k):
        if w[j] > 0:
            new[j] *= 1.0 / w[j]
        else:
            new[j] = new[amax]
    shift = np.sqrt(((new - centers) ** 2).sum(axis=1))
    return labels, new, shift


def kmeans_single(X, centers, max_iter=300, tol=1e-4):
    labels_old = np.full(X.shape[0], -1, dtype=np.int32)
    strict = False
    it = 0
    for it in range(max_iter):
        labels, new, shift = lloyd_iter(X, centers)
        centers = new
        if np.array_equal(labels, labels_old):
            strict = True
            break
        if (shift ** 2).sum() <= tol:
            break
        labels_old = labels
    if not strict:
        labels, _, _ = lloyd_iter(X, centers, update=False)
    inertia = float(((X - centers[labels]) ** 2).sum())
    return labels, inertia, centers, it + 1


def _same_clustering(l1, l2, k):
    mapping = np.full(k, -1)
    for a, b in zip(l1, l2):
        if mapping[a] == -1:
            mapping[a] = b
        elif mapping[a] != b:
            return False
    return True


def kmeans(X, k, n_init=10, random_state=1, max_iter=300, tol=1e-4):
    """KMeans(n_clusters=k, n_init=10, random_state=1).fit(X): returns (labels, centers, inertia)."""
    X = np.array(X, dtype=np.float64)
    rng = np.random.RandomState(random_state)
    tol_ = np.mean(np.var(X, axis=0)) * tol
    mean = X.mean(axis=0)
    X -= mean
    x_sq = (X * X).sum(axis=1)
    best = None
    for _ in range(n_init):
        c0, _ = kmeans_plusplus(X, k, x_sq, rng)
        labels, inertia, centers, _ = kmeans_single(X, c0, max_iter, tol_)
        if best is None or (inertia < best[1] and not _same_clustering(labels, best[0], k)):
            best = (labels, inertia, centers)
    return best[0], best[2] + mean, best[1]


def kmeans_uniforms(k, n_init=10, random_state=1):
    """The uniform doubles KMeans draws from RandomState(1), per init: 1 (first centre)
    + (k-1)*(2+int(log k)) (local trials).  The count is data independent, so a device
    implementation can take the whole stream up front."""
    rng = np.random.RandomState(random_state)
    L = 2 + int(np.log(k))
    return rng.random_sample(n_init * (1 + (k - 1) * L)).reshape(n_init, 1 + (k - 1) * L)


# ------------------------------------------------------------------ median / silhouette
def groupby_median(X, labels):
    """pandas groupby(labels).median(): groups in sorted label order; even-sized groups
    average the two middle values."""
    labs = np.unique(labels)
    return labs, np.vstack([np.median(X[labels == l], axis=0) for l in labs])


def silhouette_score(X, labels):
    D = euclidean_distances(X)
    labs, inv = np.unique(labels, return_inverse=True)
    k = labs.size
    onehot = np.zeros((X.shape[0], k))
    onehot[np.arange(X.shape[0]), inv] = 1.0
    sums = D @ onehot
    counts = onehot.sum(axis=0)
    intra = sums[np.arange(X.shape[0]), inv]
    denom = counts[inv] - 1
    with np.errstate(divide="ignore", invalid="ignore"):
        a = intra / denom
        sums[np.arange(X.shape[0]), inv] = np.inf
        b = (sums / counts[None, :]).min(axis=1)
        s = (b - a) / np.maximum(a, b)
    s = np.nan_to_num(s)
    s[denom == 0] = 0.0
    return float(np.mean(s))


# ------------------------------------------------------------------ the consensus core
def consensus_core(merged_spectra, X, k, density_threshold=0.5, local_neighborhood_size=0.30,
                   stats_mode=False, nnls_kwargs=None, refit=None):
    """cnmf.py:871-936 on arrays.  Returns a dict with every intermediate.  ``refit(X, H) -> (W, n_iter)``: the
    ``refit_usage`` solver (default: the coordinate-descent NNLS of a 'frobenius' run; a Kullback-Leibler run refits
    with ``oracle.nmf_mu.nnls_mu``, cnmf.py:618-631 + 792-798)."""
    S = np.asarray(merged_spectra, dtype=np.float64)
    n_neighbors = int(local_neighborhood_size * S.shape[0] / k)
    l2 = l2_normalise(S)
    out = {}
    if not stats_mode:
        D = euclidean_distances(l2)
        dens = local_density(D, n_neighbors)
        keep = dens < density_threshold
        out.update(topics_dist=D, local_density=dens, density_filter=keep)
        l2 = l2[keep]
        if l2.shape[0] == 0:
            raise RuntimeError("Zero components remain after density filtering. Consider increasing density threshold")
    labels0, _, inertia = kmeans(l2, k)
    labels = labels0 + 1
    labs, med = groupby_median(l2, labels)
    med = (med.T / med.sum(axis=1)).T
    kw = dict(nnls_kwargs or {})
    W, _ = (refit or nmf_cd.nnls)(np.asarray(X, dtype=np.float64), med, **kw)
    out.update(l2_spectra=l2, kmeans_labels=labels, median_spectra=med, rf_usages=W, inertia=inertia)
    if stats_mode:
        out["silhouette"] = silhouette_score(l2, labels)
        out["prediction_error"] = float(((np.asarray(X, dtype=np.float64) - W @ med) ** 2).sum())
    return out


# ------------------------------------------------------------------ the consensus tail
def ols_all_cols(Xd, Y):
    """efficient_ols_all_cols(X, Y, normalize_y=True) (cnmf.py:55-125) for dense Y:
    z-score the columns of Y (population variance, eps floor), solve the normal equations."""
    mean = Y.mean(axis=0)
    var = Y.var(axis=0)
    var[var < 1e-12] = 1e-12
    Yn = (Y - mean) / np.sqrt(var)
    beta, *_ = np.linalg.lstsq(Xd.T @ Xd, Xd.T @ Yn, rcond=None)
    return beta


def consensus_tail(core, tpm, tpm_std, hvg_idx, refit_usage=True, normalize_tpm_spectra=False, refit=None):
    """cnmf.py:939-975 on arrays: re-order programmes by total normalised usage, refit
    spectra on the TPM matrix, z-score OLS spectra, final usage refit on std-scaled HVG TPM.
    ``refit``: as in ``consensus_core``."""
    refit = refit or nmf_cd.nnls
    rf = core["rf_usages"]
    med = core["median_spectra"]
    norm = rf / rf.sum(axis=1, keepdims=True)
    order = np.argsort(-norm.sum(axis=0), kind="stable")
    rf, norm, med = rf[:, order], norm[:, order], med[order]
    tpm = np.asarray(tpm, dtype=np.float64)
    Wt, _ = refit(tpm.T, norm.T)                             # refit_spectra(tpm.X, norm_usages)
    spectra_tpm = Wt.T
    if normalize_tpm_spectra:
        spectra_tpm = spectra_tpm / spectra_tpm.sum(axis=1, keepdims=True) * 1e6
    usage_coef = ols_all_cols(rf, tpm)
    out = dict(order=order, median_spectra=med, spectra_tpm=spectra_tpm, usage_coef=usage_coef,
               rf_usages=rf)
    if refit_usage:
        norm_tpm = tpm[:, hvg_idx]
        norm_tpm = norm_tpm / norm_tpm.std(axis=0, ddof=1)
        srf = spectra_tpm[:, hvg_idx] / tpm_std[hvg_idx]
        out["rf_usages"], _ = refit(norm_tpm, srf)
    return out
