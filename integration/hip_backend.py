"""Option B of INTEGRATION.md as code: the file a maintainer drops into the reference tree as
``src/cnmf/hip_backend.py`` (or imports from here with the reference on ``sys.path``).

``cNMF`` below SUBCLASSES THE UNMODIFIED REFERENCE CLASS (``cnmf.cnmf.cNMF``): prepare, scanpy I/O, the restart
ledger, combine, plotting, ``build_reference`` and the CLI plumbing are inherited untouched; only the hot path is
overridden and sent to ``libcnmf_hip.so`` through ``cnmf_amd.engine.Engine`` (ctypes over include/cnmf_hip.h):

* ``_nmf``       (cnmf.py:661-674)  every ``non_negative_factorization`` call: restarts, ``refit_usage``,
                                    ``refit_spectra`` -- one upload per distinct matrix object;
* ``factorize``  (cnmf.py:692-745)  ONE batched device call for all ledger rows of the worker, same ``.df.npz`` files;
* ``consensus``  (cnmf.py:823-1079) the reference's own method body runs unchanged -- density cache, stats frame,
                                    re-ordering, TPM refit, OLS, file writes, clustergram, starCAT reference -- while the
                                    three scikit-learn calls inside it (``euclidean_distances`` :891/988, ``KMeans`` :908,
                                    ``silhouette_score`` :923) resolve to device-backed stand-ins for the duration of the
                                    call.

Signatures are the reference's (the CLI calls ``consensus`` positionally, cnmf.py:1290; ``factorize_mp_signature``
pickles the object, cnmf.py:254-262 -- the engine handle is dropped on pickling and re-created per process).
There is no CPU fallback: without the shared object or a GPU the overridden methods raise.
"""
import contextlib
import os

import numpy as np
import pandas as pd
import scipy.sparse as sp
import yaml

import cnmf.cnmf as _ref                      # the UNMODIFIED reference module
from cnmf.cnmf import load_df_from_npz, save_df_to_npz, worker_filter

from cnmf_amd.engine import Engine
from cnmf_amd.standins import DeviceKMeans, device_euclidean_distances, device_silhouette_score

_DEVICE_SOLVERS = {("cd", "frobenius"), ("cd", 2), ("mu", "kullback-leibler"), ("mu", "itakura-saito"), ("mu", 1), ("mu", 0)}


class cNMF(_ref.cNMF):
    _engine = None
    _untransposed = None          # set by refit_spectra around the inherited call
    _resident = None              # STRONG reference to the matrix object that is on the device (identity check)

    # -- engine plumbing ---------------------------------------------------------------------------
    def __getstate__(self):       # multiprocessing pickles the object (cnmf.py:254-262): a ctypes handle cannot travel
        d = dict(self.__dict__)
        d.pop("_engine", None)
        d.pop("_resident", None)
        d.pop("_untransposed", None)
        return d

    def _is_resident(self, X):
        """The same object, or another view of the same buffers with the same layout (``X.T.T`` of the resident matrix)."""
        R = self._resident
        if R is None or X is R:
            return R is not None
        if type(X) is not type(R) or getattr(X, "shape", None) != getattr(R, "shape", None) or X.dtype != R.dtype:
            return False
        ptr = lambda a: (a.__array_interface__["data"][0], a.shape, a.strides)          # noqa: E731
        if sp.issparse(X):
            return all(ptr(getattr(X, n)) == ptr(getattr(R, n)) for n in ("data", "indices", "indptr"))
        return isinstance(X, np.ndarray) and ptr(X) == ptr(R)

    def _eng(self, X):
        if self._engine is None:
            self._engine = Engine(int(os.environ.get("CNMF_DEVICE", "0")))
        if not self._is_resident(X):
            self._engine.set_matrix(X)
            self._resident = X
        return self._engine

    # -- cnmf.py:805-820 ---------------------------------------------------------------------------
    def refit_spectra(self, X, usage):
        """The reference's one-liner ``refit_usage(X.T, usage.T).T``, with a note for ``_nmf`` of which matrix the
        transposed view belongs to (array inputs; DataFrames take the inherited route unchanged)."""
        if isinstance(X, pd.DataFrame):
            return super().refit_spectra(X, usage)
        self._untransposed = X
        try:
            return super().refit_spectra(X, usage)
        finally:
            self._untransposed = None

    # -- cnmf.py:661-674 ---------------------------------------------------------------------------
    def _nmf(self, X, nmf_kwargs):
        kw = dict(nmf_kwargs)
        if (kw.get("solver", "cd"), kw.get("beta_loss", "frobenius")) not in _DEVICE_SOLVERS:
            raise NotImplementedError("solver=%r beta_loss=%r is not implemented on the device" % (kw.get("solver"), kw.get("beta_loss")))
        transposed = False
        U = self._untransposed
        if U is not None and kw.get("update_H", True) is False and X.shape == U.shape[::-1]:
            # refit_spectra handed over ``U.T`` (cnmf.py:820): the cells x genes matrix U itself goes to the device (the layout
            # every kernel is built for; nothing is re-uploaded if it is resident already) and the refit runs on the
            # transposed problem there
            X, transposed = U, True
        eng = self._eng(X)
        xdt = X.dtype if X.dtype in (np.float32, np.float64) else np.dtype(np.float64)
        common = dict(tol=kw.get("tol", 1e-4), max_iter=kw.get("max_iter", 200), alpha_W=kw.get("alpha_W", 0.0),
                      l1_ratio=kw.get("l1_ratio", 0.0))
        mu = kw.get("solver", "cd") == "mu"
        if kw.get("update_H", True) is False:                 # refit_usage / refit_spectra (cnmf.py:776-820)
            H = np.asarray(kw["H"])
            if H.dtype != xdt:
                raise TypeError("H should have the same dtype as X. Got H.dtype = {}.".format(H.dtype))
            # (scikit-learn solves in X's dtype: float64 matrices get the float64 device refit)
            if mu:
                # float64 on the stored entries of the matrix (cnmf_mu_refit_f64), like scikit-learn on float64 input; the
                # transposed problem walks the column-compressed image (rows = genes).  Both beta losses: an Itakura-Saito
                # run's matrices are strictly positive (scikit-learn refuses anything else), every entry is stored
                W, _, _ = eng.mu_refit_f64(H, transposed=transposed, beta_loss=kw["beta_loss"], **common)
            elif transposed:
                Hs, _ = eng.nnls_spectra(np.ascontiguousarray(H.T), **common)  # k x genes on the resident matrix
                W = np.ascontiguousarray(Hs.T)
            else:
                W, _ = (eng.nnls_f64 if xdt == np.float64 else eng.nnls)(H, **common)
            return H, W.astype(xdt, copy=False)
        k, seed = int(kw["n_components"]), int(kw["random_state"])
        if kw.get("init") == "nndsvd":
            W0, H0 = eng.nndsvd_init(k, random_state=seed)
            init = dict(W0=[W0], H0=[H0])
        else:
            init = dict(seeds=[seed])
        if mu:
            Hl, Wl, _, _ = eng.nmf_mu_batch([k], beta_loss=kw["beta_loss"], alpha_H=kw.get("alpha_H", 0.0), return_W=True,
                                            **init, **common)
        else:
            Hl, Wl, _, _ = eng.nmf_batch([k], alpha_H=kw.get("alpha_H", 0.0), return_W=True, **init, **common)
        return Hl[0].astype(xdt), Wl[0].astype(xdt)

    # -- cnmf.py:692-745 ---------------------------------------------------------------------------
    def factorize(self, worker_i=0, total_workers=1, skip_completed_runs=False):
        import scanpy as sc
        run_params = load_df_from_npz(self.paths["nmf_replicate_parameters"])
        norm_counts = sc.read(self.paths["normalized_counts"])
        kw = yaml.load(open(self.paths["nmf_run_parameters"]), Loader=yaml.FullLoader)
        if not skip_completed_runs:
            jobs = list(worker_filter(range(len(run_params)), worker_i, total_workers))
        else:
            jobs = list(worker_filter(run_params.index[run_params["completed"] == False], worker_i, total_workers))  # noqa: E712
        if not jobs:
            return
        for idx in jobs:
            print("[Worker %d]. Starting task %d." % (worker_i, idx))
        p = run_params.iloc[jobs, :]
        ks = [int(v) for v in p["n_components"].values]
        seeds = [int(v) for v in p["nmf_seed"].values]
        eng = self._eng(norm_counts.X)
        common = dict(tol=kw.get("tol", 1e-4), max_iter=kw.get("max_iter", 1000), alpha_W=kw.get("alpha_W", 0.0),
                      alpha_H=kw.get("alpha_H", 0.0), l1_ratio=kw.get("l1_ratio", 0.0))
        init = dict(seeds=seeds)
        if kw.get("init") == "nndsvd":
            inits = eng.nndsvd_init_batch(ks, seeds)           # range finders of up to 13 restarts per pass over X
            init = dict(W0=[w for w, _ in inits], H0=[h for _, h in inits])
        if kw.get("solver", "cd") == "mu":
            H, _, _, _ = eng.nmf_mu_batch(ks, beta_loss=kw["beta_loss"], **init, **common)
        else:
            H, _, _, _ = eng.nmf_batch(ks, **init, **common)              # ONE batched call: X is read per PASS, not per restart
        xdt = norm_counts.X.dtype if norm_counts.X.dtype in (np.float32, np.float64) else np.float64
        for k, it, h in zip(ks, p["iter"].values, H):                      # the files of cnmf.py:742-745
            spectra = pd.DataFrame(np.asarray(h, dtype=xdt), index=np.arange(1, k + 1), columns=norm_counts.var.index)
            save_df_to_npz(spectra, self.paths["iter_spectra"] % (k, int(it)))

    # -- cnmf.py:823-1079 --------------------------------------------------------------------------
    @contextlib.contextmanager
    def _device_sklearn(self):
        """While the reference's ``consensus`` body runs, its module-level names ``KMeans``,
        ``euclidean_distances`` and ``silhouette_score`` (cnmf.py:15-18) resolve to the device."""
        if self._engine is None:
            self._engine = Engine(int(os.environ.get("CNMF_DEVICE", "0")))
        eng, saved, last = self._engine, {}, {}

        def kmeans(n_clusters, **kw):
            last["km"] = DeviceKMeans(eng, n_clusters, **kw)
            return last["km"]

        def euclidean_distances(X, Y=None, **kw):
            return device_euclidean_distances(eng, X, Y, **kw)

        def silhouette_score(X, labels, metric="euclidean", **kw):
            return device_silhouette_score(eng, last.get("km"), X, labels, metric=metric, **kw)

        repl = dict(KMeans=kmeans, euclidean_distances=euclidean_distances, silhouette_score=silhouette_score)
        for name, fn in repl.items():
            saved[name] = getattr(_ref, name)
            setattr(_ref, name, fn)
        try:
            yield
        finally:
            for name, fn in saved.items():
                setattr(_ref, name, fn)

    def consensus(self, k, density_threshold=0.5, local_neighborhood_size=0.30, show_clustering=True,
                  build_ref=True, skip_density_and_return_after_stats=False, close_clustergram_fig=False,
                  refit_usage=True, normalize_tpm_spectra=False, norm_counts=None):
        with self._device_sklearn():
            return super().consensus(k, density_threshold, local_neighborhood_size, show_clustering, build_ref,
                                     skip_density_and_return_after_stats, close_clustergram_fig, refit_usage,
                                     normalize_tpm_spectra, norm_counts)
