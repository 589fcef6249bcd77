"""Seeded synthetic inputs for the BASELINE.json configs.

Counts come from a gamma-Poisson topic model::

    Hgt = Gamma(0.3, 1)  (K_true x G), rows normalised to sum 1
    U   = Dirichlet(0.3 * 1_K) * LogNormal(mu_lib, sigma_lib)      (N x K_true)
    C   ~ Poisson(U @ Hgt)

followed by the normalisation the reference's ``prepare`` applies to the
high-variance-gene matrix before factorisation (cnmf.py:537-544: divide every
gene column by its standard deviation with ddof=1; zero-count genes dropped,
cells with zero counts rejected at :551-554).  There is no network in the build
or GPU environment, so PBMC3k & co. are replaced by these stand-ins.
"""
import numpy as np

CONFIGS = {
    # name: (N, G, K_true, mu_lib, sigma_lib, data_seed)
    "C1": (1000, 500, 7, 7.5, 0.3, 0),      # tutorial-sized, the parity config
    "C2": (2700, 2000, 10, 6.8, 0.4, 1),    # PBMC3k stand-in
    "C3": (50000, 2000, 9, 7.5, 0.3, 2),    # north-star headline shape
    "C4": (200000, 2000, 20, 6.8, 0.4, 3),  # large sparse -> densify
}


def topic_counts(n_cells, n_genes, k_true, mu_lib=7.5, sigma_lib=0.3, seed=0, chunk=8192):
    """Integer count matrix (float32 storage) from the topic model above."""
    rs = np.random.RandomState(seed)
    Hgt = rs.gamma(0.3, 1.0, size=(k_true, n_genes))
    Hgt /= Hgt.sum(axis=1, keepdims=True)
    C = np.empty((n_cells, n_genes), dtype=np.float32)
    for s in range(0, n_cells, chunk):
        e = min(s + chunk, n_cells)
        U = rs.dirichlet(0.3 * np.ones(k_true), size=e - s)
        U *= rs.lognormal(mu_lib, sigma_lib, size=(e - s, 1))
        C[s:e] = rs.poisson(U @ Hgt)
    return C, Hgt


def normalise_like_prepare(C, dtype=np.float64):
    """Drop all-zero genes, reject zero cells, scale columns to unit variance
    (ddof=1) -- the reference's ``get_norm_counts`` dense branch, cnmf.py:540-554."""
    keep = C.sum(axis=0) > 0
    X = np.asarray(C[:, keep], dtype=dtype)
    std = X.std(axis=0, ddof=1)
    std[std == 0] = 1.0
    X = X / std
    zerocells = np.asarray(X.sum(axis=1) == 0).reshape(-1)
    if zerocells.any():
        X = X[~zerocells]
    return np.ascontiguousarray(X, dtype=dtype)


def make_config(name, dtype=np.float32, n_cells=None):
    """Normalised cells x genes matrix for one of CONFIGS (optionally truncated)."""
    N, G, K, mu, sg, seed = CONFIGS[name]
    if n_cells is not None:
        N = n_cells
    C, _ = topic_counts(N, G, K, mu, sg, seed)
    return normalise_like_prepare(C, dtype=dtype)


def consensus_stress(R=5000, G=2000, k=20, n_outliers=100, seed=0):
    """Config C5: stacked spectra for the consensus-only stress case:
    (R - n_outliers) rows = |centre + 0.05*N(0,1)| around k Gamma(0.3,1) centres,
    plus ``n_outliers`` pure-noise rows."""
    rs = np.random.RandomState(seed)
    centres = rs.gamma(0.3, 1.0, size=(k, G))
    n_good = R - n_outliers
    lab = np.arange(n_good) % k
    S = np.abs(centres[lab] + 0.05 * rs.standard_normal((n_good, G)))
    out = np.abs(rs.standard_normal((n_outliers, G)))
    S = np.vstack([S, out])
    perm = rs.permutation(R)
    return np.ascontiguousarray(S[perm]), np.concatenate([lab, -np.ones(n_outliers, int)])[perm]
