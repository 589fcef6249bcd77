"""Minimal stand-in for the parts of ``scanpy`` the reference imports (TEST ORACLE).

scanpy / anndata / h5py are absent from this image and there is no network, so
``import cnmf`` fails as shipped (cnmf.py:26).  The hot path only needs
``sc.AnnData``, ``sc.read``, ``sc.write``, ``sc.pp.normalize_total`` and ``sc.pp.scale``
(SURVEY.md appendix B).  ``install()`` puts this module into ``sys.modules['scanpy']`` and
``/root/reference/src`` on ``sys.path`` so that the UNMODIFIED reference runs
prepare -> factorize -> combine -> consensus -> k_selection_plot in the build container.
Used only by tools/make_golden.py and the build-container-only tests; never on the GPU box
(the reference tree does not exist there).

Files are pickles regardless of the ``.h5ad`` extension (a round-trippable container is all
the reference needs, cnmf.py:384,410,561,726,873,950).
"""
import pickle
import sys
import types

import numpy as np
import pandas as pd
import scipy.sparse as sp

REFERENCE_SRC = "/root/reference/src"


class AnnData:
    def __init__(self, X=None, obs=None, var=None):
        self.X = X
        n, g = X.shape
        self.obs = obs if obs is not None else pd.DataFrame(index=[str(i) for i in range(n)])
        self.var = var if var is not None else pd.DataFrame(index=[str(i) for i in range(g)])

    @property
    def shape(self):
        return self.X.shape

    @property
    def obs_names(self):
        return self.obs.index

    @property
    def var_names(self):
        return self.var.index

    def copy(self):
        return AnnData(self.X.copy(), self.obs.copy(), self.var.copy())

    def __getitem__(self, key):
        rows, cols = key if isinstance(key, tuple) else (key, slice(None))
        ci = (np.arange(self.shape[1])[cols] if isinstance(cols, slice)
              else self.var.index.get_indexer(list(cols)))
        ri = (np.arange(self.shape[0])[rows] if isinstance(rows, slice)
              else self.obs.index.get_indexer(list(rows)))
        X = self.X[ri][:, ci]
        return AnnData(X, self.obs.iloc[ri].copy(), self.var.iloc[ci].copy())


def read(fn, **kw):
    with open(fn, "rb") as f:
        return pickle.load(f)


def write(fn, adata, **kw):
    with open(fn, "wb") as f:
        pickle.dump(adata, f, protocol=4)


def _normalize_total(adata, target_sum=None, copy=False, **kw):
    X = adata.X
    if sp.issparse(X):
        counts = np.asarray(X.sum(axis=1)).ravel()
        scale = np.where(counts > 0, target_sum / np.where(counts > 0, counts, 1), 0.0)
        adata.X = sp.diags(scale) @ X.astype(np.float64)
        adata.X = sp.csr_matrix(adata.X)
    else:
        counts = X.sum(axis=1, keepdims=True)
        adata.X = X.astype(np.float64) / np.where(counts > 0, counts, 1) * target_sum
    return adata if copy else None


def _scale(adata, zero_center=False, **kw):
    """Documented behaviour of sc.pp.scale(zero_center=False): divide every column by its
    standard deviation (ddof=1); zero-variance columns are left unscaled."""
    X = adata.X
    if sp.issparse(X):
        X = X.tocsc().astype(np.float64)
        n = X.shape[0]
        mean = np.asarray(X.mean(axis=0)).ravel()
        sq = np.asarray(X.multiply(X).mean(axis=0)).ravel()
        var = (sq - mean ** 2) * n / (n - 1)
        std = np.sqrt(var)
        std[std == 0] = 1
        adata.X = sp.csr_matrix(X @ sp.diags(1.0 / std))
    else:
        std = X.std(axis=0, ddof=1)
        std[std == 0] = 1
        adata.X = X / std


def install():
    """Register the shim as ``scanpy`` and make ``import cnmf`` resolve to the reference."""
    mod = types.ModuleType("scanpy")
    mod.AnnData = AnnData
    mod.read = read
    mod.write = write
    mod.read_h5ad = read
    mod.pp = types.SimpleNamespace(normalize_total=_normalize_total, scale=_scale)
    sys.modules["scanpy"] = mod
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
    import matplotlib
    matplotlib.use("Agg")
    return mod
