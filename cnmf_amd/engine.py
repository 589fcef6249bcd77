"""Python face of the C-ABI: one ``Engine`` = one ``cnmf_ctx`` = one MI355X.

The engine keeps the cells x genes matrix resident in HBM and runs batches of
NMF restarts / NNLS refits on it.  Everything numeric happens in
``libcnmf_hip.so``; this module only marshals numpy buffers through ctypes and
maps status codes to the exception types the reference path raises
(scikit-learn raises ValueError / TypeError).
"""
import ctypes as C
import os
import warnings

import numpy as np

from . import _lib

_ERR = {-1: ValueError, -2: RuntimeError, -3: MemoryError, -4: RuntimeError,
        -5: NotImplementedError, -6: RuntimeError}


class ConvergenceWarning(UserWarning):
    """Mirror of sklearn.exceptions.ConvergenceWarning (sklearn _nmf.py:1727-1732)."""


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def regularization(n_samples, n_features, alpha_W=0.0, alpha_H=0.0, l1_ratio=0.0):
    """sklearn's scaling of the penalties, decomposition/_nmf.py:1254-1265."""
    return (n_features * alpha_W * l1_ratio, n_samples * alpha_H * l1_ratio,
            n_features * alpha_W * (1.0 - l1_ratio), n_samples * alpha_H * (1.0 - l1_ratio))


def _nndsvd_finish(k, Q, B, transpose, eps=1e-6):
    """The host tail of one NNDSVD initialisation (sklearn utils/extmath.py:588-602 + decomposition/_nmf.py:317-354):
    Q [M_rows, c] orthonormal, B = Q^T M [c, M_cols] -> small SVD, svd_flip, positive / negative split, threshold.
    Everything runs on ROWS of length n_samples / n_features (the singular vectors are kept transposed: k x M_rows from
    ONE product Uhat[:, :k]^T Q^T, only the k vectors that are used) -- the column-wise form of the same arithmetic spent
    20 ms of its 40 in one strided arg-max."""
    from scipy import linalg
    Q = np.asarray(Q, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64)
    Uhat, s, Vt = linalg.svd(B, full_matrices=False, lapack_driver="gesdd")
    UT = Uhat[:, :k].T @ Q.T                                      # [k, M_rows] = (Q Uhat)[:, :k]^T
    VT = Vt[:k, :]                                                # [k, M_cols]
    # svd_flip: u_based_decision on the matrix that was decomposed (M = X, or X^T when n_samples < n_features, where
    # sklearn transposes back BEFORE flipping: the decision is then taken on the rows of Vt)
    lead = VT if transpose else UT
    idx = np.argmax(np.abs(lead), axis=1)
    signs = np.sign(lead[np.arange(k), idx])
    UT = UT * signs[:, np.newaxis]
    VT = VT * signs[:, np.newaxis]
    WT, HR = (VT, UT) if transpose else (UT, VT)                  # rows: left vectors over samples, right over features
    S = s[:k]
    W = np.zeros_like(WT)
    H = np.zeros_like(HR)
    W[0] = np.sqrt(S[0]) * np.abs(WT[0])
    H[0] = np.sqrt(S[0]) * np.abs(HR[0])
    for j in range(1, k):
        x, y = WT[j], HR[j]
        x_p, y_p = np.maximum(x, 0), np.maximum(y, 0)
        x_n, y_n = np.abs(np.minimum(x, 0)), np.abs(np.minimum(y, 0))
        x_p_nrm, y_p_nrm = np.sqrt(x_p @ x_p), np.sqrt(y_p @ y_p)
        x_n_nrm, y_n_nrm = np.sqrt(x_n @ x_n), np.sqrt(y_n @ y_n)
        m_p, m_n = x_p_nrm * y_p_nrm, x_n_nrm * y_n_nrm
        if m_p > m_n:
            u, v, sigma = x_p / x_p_nrm, y_p / y_p_nrm, m_p
        else:
            u, v, sigma = x_n / x_n_nrm, y_n / y_n_nrm, m_n
        lbd = np.sqrt(S[j] * sigma)
        W[j] = lbd * u
        H[j] = lbd * v
    W[W < eps] = 0
    H[H < eps] = 0
    return np.ascontiguousarray(W.T), H


class Engine:
    def __init__(self, device=0, detect_counts=True):
        """``detect_counts=False``: never take the integer-plane (count-structured) GEMM path -- see
        :meth:`set_count_detection`."""
        self._lib = _lib.load()
        if self._lib.cnmf_device_count() <= 0:
            raise RuntimeError("cnmf_amd: no HIP device visible -- the engine has no CPU fallback")
        self._ctx = self._lib.cnmf_create(int(device))
        if not self._ctx:
            raise RuntimeError("cnmf_create failed: %s" % self._lib.cnmf_last_error(None).decode())
        self.device = int(device)
        self._env_snap = self._env_now()
        self.shape = None
        self._x_mean, self._x_mean_src = None, None
        self.x_dtype = None
        self.x_has_zero = False            # (scikit-learn refuses beta_loss <= 0 on a matrix that contains a zero)
        self.last_stats = None
        self.store_gen = 0                 # generation of the resident spectra store (bumped by spectra_reset)
        self.last_store_offsets = None     # first store row of every restart of the last resident batch
        self.last_store_gen = -1
        if not detect_counts:
            self.set_count_detection(False)

    def set_count_detection(self, enabled):
        """The engine recognises ``X = counts / std`` (integers x one constant per gene, tolerance 1e-3 count units;
        DESIGN.md section 4) and then multiplies exact integer planes.  A matrix that merely happens to lie that close
        to such a grid would be snapped onto it: switch the detection off for data that is not count-derived
        (same effect as the environment variable ``CNMF_NO_COUNTS=1``)."""
        self._check(self._lib.cnmf_set_count_detection(self._ctx, int(bool(enabled))))

    # ------------------------------------------------------------------ plumbing
    @staticmethod
    def _env_now():
        return tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith("CNMF_")))

    def reload_env(self):
        """The library reads its ``CNMF_*`` switches once per context (cnmf_create); this re-reads them."""
        self._check(self._lib.cnmf_reload_env(self._ctx))
        self._env_snap = self._env_now()

    def _sync_env(self):
        # host-side convenience (tests and A/B tools flip a switch between two calls of one process): the context's
        # snapshot follows os.environ when -- and only when -- the process environment has changed since it was taken
        if self._env_now() != self._env_snap:
            self.reload_env()

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.cnmf_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != 0:
            msg = self._lib.cnmf_last_error(self._ctx).decode()
            raise _ERR.get(rc, RuntimeError)("cnmf_hip: %s (code %d)" % (msg, rc))

    def _params(self, tol, max_iter, alpha_W, alpha_H, l1_ratio, kc_max=0, lag=0, profile=0, n_features=None):
        """``n_features``: the feature count sklearn would see (it scales the W penalties, _nmf.py:1254-1265) when the
        solved problem uses a column SUBSET of the resident matrix (nnls_gram)."""
        self._sync_env()
        N, G = self.shape
        l1W, l1H, l2W, l2H = regularization(N, G if n_features is None else int(n_features), alpha_W, alpha_H, l1_ratio)
        return _lib.CdParams(float(tol), int(max_iter), int(kc_max), l1W, l2W, l1H, l2H, int(lag), int(profile))

    # ------------------------------------------------------------------ data matrix
    def set_matrix(self, X):
        """Upload the cells x genes matrix (dense ndarray or scipy CSR), once per context.

        Mirrors sklearn's input validation for NMF (check_array + check_non_negative,
        decomposition/_nmf.py:1126,283): NaN/inf and negative entries raise ValueError."""
        import scipy.sparse as sp
        if sp.issparse(X):
            X = X.tocsr()
            if not X.has_canonical_format:          # (sorted rows without duplicates: the device keeps the arrays as uploaded)
                X = X.copy()
                X.sum_duplicates()
            data = X.data
            if not np.isfinite(data).all():
                raise ValueError("Input X contains NaN or infinity.")
            if data.size and data.min() < 0:
                raise ValueError("Negative values in data passed to NMF (input X)")
            self.x_dtype = np.dtype(X.dtype) if X.dtype in (np.float32, np.float64) else np.dtype(np.float64)
            # scipy's sparse mean (what scikit-learn's init='random' divides by) copies, divides and sums the whole matrix:
            # 0.17 s at 50 000 x 2 000 -- taken only when a restart asks for it (the TPM upload of consensus() never does)
            self._x_mean, self._x_mean_src = None, X
            self.x_has_zero = bool(X.nnz < X.shape[0] * X.shape[1] or (data.size and data.min() == 0))
            vals = np.ascontiguousarray(data, dtype=np.float32)
            if vals.size and not vals.all():
                # a STORED zero (explicit, or a float64 value that underflows in float32) would make the library treat the
                # arrays as non-canonical and form the dense image of the whole matrix (round-5 advice): compact them here
                Xc = sp.csr_matrix((vals, X.indices.copy(), X.indptr.copy()), shape=X.shape)    # (eliminate_zeros works in place:
                Xc.eliminate_zeros()                                                            #  never on the caller's arrays)
                X, vals = Xc, np.ascontiguousarray(Xc.data, dtype=np.float32)
            indptr = np.ascontiguousarray(X.indptr, dtype=np.int32)
            indices = np.ascontiguousarray(X.indices, dtype=np.int32)
            ip = C.POINTER(C.c_int32)
            self._check(self._lib.cnmf_set_matrix_csr(self._ctx, indptr.ctypes.data_as(ip),
                                                      indices.ctypes.data_as(ip), _fp(vals),
                                                      X.shape[0], X.shape[1]))
        else:
            X = np.asarray(X)
            if X.ndim != 2:
                raise ValueError("Expected 2D array, got %dD array instead" % X.ndim)
            if X.dtype not in (np.float32, np.float64):
                X = X.astype(np.float64)            # sklearn: ints -> float64
            if not np.isfinite(X).all():
                raise ValueError("Input X contains NaN or infinity.")
            if X.size and X.min() < 0:
                raise ValueError("Negative values in data passed to NMF (input X)")
            self.x_dtype = X.dtype
            self.x_has_zero = bool(X.size and X.min() == 0)
            self.x_mean = X.mean()                  # numpy scalar of X's dtype, like sklearn's X.mean()
            Xf = np.ascontiguousarray(X, dtype=np.float32)
            self._check(self._lib.cnmf_set_matrix(self._ctx, _fp(Xf), Xf.shape[0], Xf.shape[1]))
        self.shape = (int(X.shape[0]), int(X.shape[1]))
        return self

    def matrix_images(self):
        """Which images of the matrix the device holds right now (cnmf_matrix_images): a CSR upload keeps its compressed
        rows and forms the dense float32 image only when a path that multiplies the dense matrix asks for it."""
        f = C.c_int32(0)
        self._check(self._lib.cnmf_matrix_images(self._ctx, C.byref(f)))
        names = ("dense", "csr", "csr_of_transpose", "dense_transpose", "non_zero_images_16", "non_zero_images_32", "count_planes")
        return {n: bool(f.value >> i & 1) for i, n in enumerate(names)}

    def get_matrix(self):
        """The resident matrix back on the host (float32 [n_cells, n_genes])."""
        if self.shape is None:
            raise RuntimeError("set_matrix() has not been called")
        out = np.empty(self.shape, dtype=np.float32)
        self._check(self._lib.cnmf_get_matrix(self._ctx, _fp(out)))
        return out

    def col_mean_var(self):
        """Per-gene mean and POPULATION variance (ddof=0) of the resident matrix in float64 -- the
        O(cells x genes) part of the reference's high-variance-gene statistics (cnmf.py:195-196, 126-129)."""
        if self.shape is None:
            raise RuntimeError("set_matrix() has not been called")
        N, G = self.shape
        dblp = C.POINTER(C.c_double)
        mean, ssd = np.empty(G), np.empty(G)
        self._check(self._lib.cnmf_col_moments(self._ctx, mean.ctypes.data_as(dblp), ssd.ctypes.data_as(dblp)))
        return mean, ssd / N

    def scale_genes_unit_variance(self):
        """``X /= X.std(axis=0, ddof=1)`` on the resident matrix -- the dense branch of the reference's
        ``get_norm_counts`` (cnmf.py:540-548), statistics in float64.  Returns ``(std, row_sums)``;
        ``row_sums == 0`` marks the reference's "zero cells" (cnmf.py:550-554).  Afterwards the matrix
        counts as float64 input, like ``norm_counts.X`` in the reference (cnmf.py:534)."""
        if self.shape is None:
            raise RuntimeError("set_matrix() has not been called")
        N, G = self.shape
        if N < 2:
            raise ValueError("need at least two cells for a variance")
        dblp = C.POINTER(C.c_double)
        mean, ssd = np.empty(G), np.empty(G)
        self._check(self._lib.cnmf_col_moments(self._ctx, mean.ctypes.data_as(dblp), ssd.ctypes.data_as(dblp)))
        std = np.sqrt(ssd / (N - 1))
        self._check(self._lib.cnmf_scale_columns(self._ctx, std.ctypes.data_as(dblp)))
        rs = np.empty(N)
        self._check(self._lib.cnmf_row_sums(self._ctx, rs.ctypes.data_as(dblp)))
        self.x_dtype = np.dtype(np.float64)
        self.x_mean = np.float64(rs.sum() / (float(N) * float(G)))
        return std, rs

    @property
    def x_mean(self):
        if self._x_mean is None and self._x_mean_src is not None:
            X = self._x_mean_src
            self._x_mean = X.mean() if X.dtype in (np.float32, np.float64) else X.astype(np.float64).mean()
            self._x_mean_src = None
        return self._x_mean

    @x_mean.setter
    def x_mean(self, v):
        self._x_mean, self._x_mean_src = v, None

    def init_scale(self, k):
        """``avg = sqrt(X.mean() / n_components)`` exactly as sklearn computes it
        (numpy scalar arithmetic in X's dtype; decomposition/_nmf.py:303)."""
        return float(np.sqrt(self.x_mean / k))

    # ------------------------------------------------------------------ restarts
    def nmf_batch(self, ks, seeds=None, W0=None, H0=None, tol=1e-4, max_iter=1000,
                  alpha_W=0.0, alpha_H=0.0, l1_ratio=0.0, return_W=False, kc_max=0, lag=0,
                  resident=False, warn=True, profile=False):
        """Run ``len(ks)`` independent CD-NMF restarts on the resident matrix.

        Either ``seeds`` (sklearn ``init='random'`` reproduced on the device) or the lists
        ``W0`` / ``H0`` of explicit initial factors (sklearn layouts N x k and k x G).
        Returns ``(H_list, W_list_or_None, n_iter, violation)``."""
        if self.shape is None:
            raise RuntimeError("set_matrix() has not been called")
        N, G = self.shape
        ks = np.ascontiguousarray(ks, dtype=np.int32).ravel()
        n = int(ks.size)
        if n and ks.min() < 1:
            raise ValueError("n_components must be >= 1")
        prm = self._params(tol, max_iter, alpha_W, alpha_H, l1_ratio, kc_max, lag, profile)
        i32p, u32p, dblp = C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_double)
        if seeds is not None:
            seeds_a = np.ascontiguousarray(np.asarray(seeds, dtype=np.int64).astype(np.uint32)).ravel()
            if seeds_a.size != n:
                raise ValueError("need one seed per restart")
            avg = np.ascontiguousarray([self.init_scale(int(k)) for k in ks], dtype=np.float64)
            mode, w0p, h0p = 1, None, None
            seedp, avgp = seeds_a.ctypes.data_as(u32p), avg.ctypes.data_as(dblp)
        else:
            if W0 is None or H0 is None or len(W0) != n or len(H0) != n:
                raise ValueError("need seeds, or W0 and H0 lists with one entry per restart")
            for r in range(n):
                if np.shape(W0[r]) != (N, ks[r]) or np.shape(H0[r]) != (ks[r], G):
                    raise ValueError("Array with wrong shape passed to NMF (input W/H) for restart %d" % r)
                if np.min(W0[r]) < 0 or np.min(H0[r]) < 0:
                    raise ValueError("Negative values in data passed to NMF (input W/H)")
            w0 = (np.concatenate([np.ascontiguousarray(w, dtype=np.float32).ravel() for w in W0])
                  if n else np.zeros(1, np.float32))
            h0 = (np.concatenate([np.ascontiguousarray(h, dtype=np.float32).ravel() for h in H0])
                  if n else np.zeros(1, np.float32))
            mode, w0p, h0p, seedp, avgp = 0, _fp(w0), _fp(h0), None, None
        tot_k = int(ks.sum()) if n else 0
        n_iter = np.zeros(max(n, 1), dtype=np.int32)
        viol = np.zeros(max(n, 1), dtype=np.float64)
        stats = _lib.BatchStats()
        if resident:
            # the spectra stay in the context's device store (appended behind what it holds); resident="keep" also
            # brings THIS call's rows to the host (ONE copy), and `last_store_offsets` says where each restart sits in the
            # store -- consensus() / kselect_stats() can then take their merged spectra from the device (store_rows)
            row0 = self.spectra_rows
            rc = self._lib.cnmf_nmf_cd_batch_resident(self._ctx, n, ks.ctypes.data_as(i32p), mode, seedp,
                                                      avgp, w0p, h0p, C.byref(prm),
                                                      n_iter.ctypes.data_as(i32p), viol.ctypes.data_as(dblp),
                                                      C.byref(stats))
            self._check(rc)
            H_list = W_list = None
            offs = row0 + np.concatenate([[0], np.cumsum(ks)]).astype(np.int64)
            self.last_store_offsets = offs[:-1].copy()
            self.last_store_gen = self.store_gen
            if resident == "keep":
                if return_W:
                    raise ValueError("resident='keep' returns spectra only")
                mine = self.spectra_fetch(row0, tot_k)            # THIS call's rows only (ONE copy)
                H_list = [mine[offs[r] - row0:offs[r + 1] - row0] for r in range(n)]
        else:
            H_out = np.empty((max(tot_k, 1), G), dtype=np.float32)
            W_out = np.empty(max(tot_k, 1) * N, dtype=np.float32) if return_W else None
            rc = self._lib.cnmf_nmf_cd_batch(self._ctx, n, ks.ctypes.data_as(i32p), mode, seedp, avgp,
                                             w0p, h0p, C.byref(prm), _fp(H_out),
                                             _fp(W_out) if return_W else None,
                                             n_iter.ctypes.data_as(i32p), viol.ctypes.data_as(dblp),
                                             C.byref(stats))
            self._check(rc)
            offs = np.concatenate([[0], np.cumsum(ks)]).astype(np.int64)
            H_list = [H_out[offs[r]:offs[r + 1]] for r in range(n)]
            W_list = None
            if return_W:
                W_list = [W_out[offs[r] * N:offs[r + 1] * N].reshape(N, ks[r]) for r in range(n)]
        self.last_stats = stats.as_dict()
        n_iter, viol = n_iter[:n], viol[:n]
        if warn and n and tol > 0 and (n_iter == max_iter).any():
            warnings.warn("Maximum number of iterations %d reached. Increase it to improve convergence."
                          % max_iter, ConvergenceWarning)
        return H_list, W_list, n_iter, viol

    def iteration_means(self):
        """``{rank: mean outer iterations}`` of the restarts the batch calls on the resident matrix have run so far."""
        out = np.zeros(_lib.CNMF_KMAX + 1, dtype=np.float64)
        self._check(self._lib.cnmf_get_iteration_means(self._ctx, out.ctypes.data_as(C.POINTER(C.c_double))))
        return {int(k): float(v) for k, v in enumerate(out) if v > 0}

    def set_iteration_hints(self, hints=None):
        """Expected iterations per rank (e.g. :meth:`iteration_means` of an earlier call) for the queue order of the following
        ``nmf_batch`` calls on this matrix -- longest-expected-first from the start instead of learning the order again;
        ``None`` clears them.  Explicit by design: the order decides which packed columns a restart occupies, and its
        float32 result moves in the last bits with its placement."""
        hints = hints or {}
        ks = np.ascontiguousarray(sorted(hints), dtype=np.int32)
        vals = np.ascontiguousarray([hints[int(k)] for k in ks], dtype=np.float64)
        self._check(self._lib.cnmf_set_iteration_hints(self._ctx, int(ks.size), ks.ctypes.data_as(C.POINTER(C.c_int32)),
                                                       vals.ctypes.data_as(C.POINTER(C.c_double))))

    # ------------------------------------------------------------------ multiplicative update
    _BETA = {"kullback-leibler": 1, "itakura-saito": 0, 1: 1, 0: 0, 1.0: 1, 0.0: 0}

    def nmf_mu_batch(self, ks, seeds=None, W0=None, H0=None, beta_loss="kullback-leibler", tol=1e-4,
                     max_iter=1000, alpha_W=0.0, alpha_H=0.0, l1_ratio=0.0, return_W=False, warn=True):
        """``solver='mu'`` restarts (the reference's path for beta_loss != 'frobenius',
        cnmf.py:618-631).  Same arguments / returns as :meth:`nmf_batch`; the last element of the
        returned tuple is sqrt(2*beta-divergence) at the final evaluation."""
        if self.shape is None:
            raise RuntimeError("set_matrix() has not been called")
        if beta_loss not in self._BETA:
            raise NotImplementedError("beta_loss=%r is not implemented on the device" % (beta_loss,))
        self._refuse_zeros(beta_loss)
        N, G = self.shape
        ks = np.ascontiguousarray(ks, dtype=np.int32).ravel()
        n = int(ks.size)
        prm = self._params(tol, max_iter, alpha_W, alpha_H, l1_ratio)
        i32p, u32p, dblp = C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_double)
        avg = np.ascontiguousarray([self.init_scale(int(k)) for k in ks], dtype=np.float64)
        if seeds is not None:
            seeds_a = np.ascontiguousarray(np.asarray(seeds, dtype=np.int64).astype(np.uint32)).ravel()
            mode, w0p, h0p, seedp = 1, None, None, seeds_a.ctypes.data_as(u32p)
        else:
            w0 = np.concatenate([np.ascontiguousarray(w, dtype=np.float32).ravel() for w in W0])
            h0 = np.concatenate([np.ascontiguousarray(h, dtype=np.float32).ravel() for h in H0])
            mode, w0p, h0p, seedp = 0, _fp(w0), _fp(h0), None
        tot_k = int(ks.sum()) if n else 0
        H_out = np.empty((max(tot_k, 1), G), dtype=np.float32)
        W_out = np.empty(max(tot_k, 1) * N, dtype=np.float32) if return_W else None
        n_iter = np.zeros(max(n, 1), dtype=np.int32)
        err = np.zeros(max(n, 1), dtype=np.float64)
        self._check(self._lib.cnmf_nmf_mu_batch(self._ctx, n, ks.ctypes.data_as(i32p), mode, seedp,
                                                avg.ctypes.data_as(dblp), w0p, h0p, self._BETA[beta_loss], 1,
                                                C.byref(prm), _fp(H_out), _fp(W_out) if return_W else None,
                                                n_iter.ctypes.data_as(i32p), err.ctypes.data_as(dblp)))
        offs = np.concatenate([[0], np.cumsum(ks)]).astype(np.int64)
        H_list = [H_out[offs[r]:offs[r + 1]] for r in range(n)]
        W_list = [W_out[offs[r] * N:offs[r + 1] * N].reshape(N, ks[r]) for r in range(n)] if return_W else None
        n_iter, err = n_iter[:n], err[:n]
        if warn and n and tol > 0 and (n_iter == max_iter).any():
            warnings.warn("Maximum number of iterations %d reached. Increase it to improve convergence."
                          % max_iter, ConvergenceWarning)
        return H_list, W_list, n_iter, err

    def nnls_mu(self, H, beta_loss="kullback-leibler", tol=1e-4, max_iter=1000, alpha_W=0.0, l1_ratio=0.0,
                warn=True):
        """Refit with fixed H and ``solver='mu'``: W starts from avg everywhere (sklearn _nmf.py:1229-1231).
        Both losses: the float64 refit on the stored entries (:meth:`mu_refit_f64`; result cast to float32 like the
        other refits of this class)."""
        if self.shape is None:
            raise RuntimeError("set_matrix() has not been called")
        N, G = self.shape
        H = np.asarray(H)
        if H.ndim != 2 or H.shape[1] != G:
            raise ValueError("Array with wrong shape passed to NMF (input H).")
        if H.min() < 0:
            raise ValueError("Negative values in data passed to NMF (input H)")
        if H.max() == 0:
            raise ValueError("Array passed to NMF (input H) is full of zeros.")
        W, n, _ = self.mu_refit_f64(H, tol=tol, max_iter=max_iter, alpha_W=alpha_W, l1_ratio=l1_ratio, warn=warn,
                                    beta_loss=beta_loss)
        return W.astype(np.float32), n

    _ZERO_MSG = ("When beta_loss <= 0 and X contains zeros, the solver may diverge. Please add small values to X, or use a "
                 "positive beta_loss.")

    def _refuse_zeros(self, beta_loss):
        """scikit-learn's own refusal (``_fit_transform``, _nmf.py:1679-1684): ``beta_loss <= 0`` on a matrix that contains a
        zero raises ValueError -- what the reference's ``factorize`` / ``refit_usage`` do for every ordinary count matrix under
        ``--beta-loss itakura-saito``."""
        if self._BETA[beta_loss] <= 0 and self.x_has_zero:
            raise ValueError(self._ZERO_MSG)

    def mu_refit_f64(self, H, transposed=False, col_divisor=None, w_init=None, tol=1e-4, max_iter=1000, alpha_W=0.0,
                     l1_ratio=0.0, n_features=None, warn=True, beta_loss="kullback-leibler"):
        """``non_negative_factorization(X, H=H, update_H=False, solver='mu', beta_loss=...)`` in float64 on
        the stored entries of the resident matrix (cnmf_mu_refit_f64): returns ``(W float64, n_iter, err)``.
        ``beta_loss='itakura-saito'`` (round 6): the matrix must be strictly positive (scikit-learn's rule), so every entry
        is a stored entry.

        ``transposed=False``: X = the resident cells x genes matrix, ``H`` [k, n_genes], W [n_cells, k] (refit_usage,
        cnmf.py:776-802).  ``transposed=True``: the problem on X^T -- ``H`` [k, n_cells] (usages^T), W [n_genes, k]
        (refit_spectra, cnmf.py:805-820) -- on the compressed rows of X^T built on the device.
        ``col_divisor`` [columns of the walked matrix]: the matrix meant is ``X[:, d != 0] / d[d != 0]`` (the final usage
        refit on the unit-variance high-variance-gene TPM, cnmf.py:963-972); then ``w_init`` (scikit-learn's
        sqrt(mean / k) of THAT matrix) must be given and ``n_features`` is its column count (scales the W penalties)."""
        self._sync_env()
        if self.shape is None:
            raise RuntimeError("set_matrix() has not been called")
        if beta_loss not in self._BETA:
            raise NotImplementedError("beta_loss=%r is not implemented on the device" % (beta_loss,))
        self._refuse_zeros(beta_loss)
        N, G = self.shape
        rows, cols = (G, N) if transposed else (N, G)
        H = np.ascontiguousarray(H, dtype=np.float64)
        if H.ndim != 2 or H.shape[1] != cols:
            raise ValueError("Array with wrong shape passed to NMF (input H). Expected (k, %d), but got %s"
                             % (cols, (H.shape,)))
        if not np.isfinite(H).all():
            raise ValueError("Input H contains NaN or infinity.")
        if H.min() < 0:
            raise ValueError("Negative values in data passed to NMF (input H)")
        if H.max() == 0:
            raise ValueError("Array passed to NMF (input H) is full of zeros.")
        k = int(H.shape[0])
        dblp = C.POINTER(C.c_double)
        div = None
        if col_divisor is not None:
            div = np.ascontiguousarray(col_divisor, dtype=np.float64)
            if div.shape != (cols,) or w_init is None:
                raise ValueError("col_divisor needs one entry per column of the walked matrix and an explicit w_init")
        if w_init is None:
            w_init = self.init_scale(k)
        # scikit-learn scales the W penalties by the FEATURE count of the matrix it is given (_nmf.py:1254-1265)
        feats = cols if n_features is None else int(n_features)
        l1W, _, l2W, _ = regularization(rows, feats, alpha_W, 0.0, l1_ratio)
        prm = _lib.CdParams(float(tol), int(max_iter), 0, l1W, l2W, 0.0, 0.0, 0, 0)
        W = np.empty((rows, k), dtype=np.float64)
        n_iter = C.c_int32(0)
        err = C.c_double(0.0)
        self._check(self._lib.cnmf_mu_refit_f64(self._ctx, 1 if transposed else 0, self._BETA[beta_loss], k, H.ctypes.data_as(dblp),
                                                div.ctypes.data_as(dblp) if div is not None else None, float(w_init),
                                                C.byref(prm), W.ctypes.data_as(dblp), C.byref(n_iter), C.byref(err)))
        if warn and tol > 0 and n_iter.value == max_iter:
            warnings.warn("Maximum number of iterations %d reached. Increase it to improve convergence."
                          % max_iter, ConvergenceWarning)
        return W, int(n_iter.value), float(err.value)

    # ------------------------------------------------------------------ NNLS refit
    def nnls(self, H, tol=1e-4, max_iter=1000, alpha_W=0.0, l1_ratio=0.0, warn=True):
        """``non_negative_factorization(X, H=H, update_H=False, solver='cd')`` on the
        resident matrix: returns ``(W [N x k], n_iter)``."""
        if self.shape is None:
            raise RuntimeError("set_matrix() has not been called")
        N, G = self.shape
        H = np.asarray(H)
        if H.ndim != 2 or H.shape[1] != G:
            raise ValueError("Array with wrong shape passed to NMF (input H). Expected (k, %d), but got %s"
                             % (G, (H.shape,)))
        if not np.isfinite(H).all():
            raise ValueError("Input H contains NaN or infinity.")
        if H.min() < 0:
            raise ValueError("Negative values in data passed to NMF (input H)")
        if H.max() == 0:
            raise ValueError("Array passed to NMF (input H) is full of zeros.")
        k = int(H.shape[0])
        Hf = np.ascontiguousarray(H, dtype=np.float32)
        prm = self._params(tol, max_iter, alpha_W, 0.0, l1_ratio)
        W = np.empty((N, k), dtype=np.float32)
        n_iter = C.c_int32(0)
        viol = C.c_double(0.0)
        self._check(self._lib.cnmf_nnls(self._ctx, k, _fp(Hf), C.byref(prm), _fp(W),
                                        C.byref(n_iter), C.byref(viol)))
        if warn and tol > 0 and n_iter.value == max_iter:
            warnings.warn("Maximum number of iterations %d reached. Increase it to improve convergence."
                          % max_iter, ConvergenceWarning)
        return W, int(n_iter.value)

    # ------------------------------------------------------------------ consensus tail + k selection
    def _check_H(self, H, G):
        H = np.asarray(H)
        if H.ndim != 2 or H.shape[1] != G:
            raise ValueError("Array with wrong shape passed to NMF (input H). Expected (k, %d), but got %s"
                             % (G, (H.shape,)))
        if not np.isfinite(H).all():
            raise ValueError("Input H contains NaN or infinity.")
        if H.min() < 0:
            raise ValueError("Negative values in data passed to NMF (input H)")
        if H.max() == 0:
            raise ValueError("Array passed to NMF (input H) is full of zeros.")
        return H

    def nnls_batch(self, H_list, tol=1e-4, max_iter=1000, alpha_W=0.0, l1_ratio=0.0, return_W=True,
                   prediction_error=False):
        """Several usage refits (``cNMF.refit_usage``, cnmf.py:776-802) as ONE pass over the resident matrix:
        every ``H`` of ``H_list`` (k_r x G) becomes columns of a single X.H^T product, the W sweeps run together.
        Returns ``(W_list or None, n_iter, errors or None)`` -- ``errors[r] = ||X - W_r H_r||_F^2`` in float64."""
        if self.shape is None:
            raise RuntimeError("set_matrix() has not been called")
        N, G = self.shape
        Hs = [np.ascontiguousarray(self._check_H(H, G), dtype=np.float32) for H in H_list]
        n = len(Hs)
        ks = np.ascontiguousarray([h.shape[0] for h in Hs], dtype=np.int32)
        if n == 0:
            return ([] if return_W else None), np.zeros(0, np.int32), (np.zeros(0) if prediction_error else None)
        Hp = np.ascontiguousarray(np.concatenate(Hs, axis=0))
        prm = self._params(tol, max_iter, alpha_W, 0.0, l1_ratio)
        W_out = np.empty(int(ks.sum()) * N, dtype=np.float32) if return_W else None
        n_iter = np.zeros(n, dtype=np.int32)
        viol = np.zeros(n, dtype=np.float64)
        err = np.zeros(n, dtype=np.float64) if prediction_error else None
        i32p, dblp = C.POINTER(C.c_int32), C.POINTER(C.c_double)
        self._check(self._lib.cnmf_nnls_batch(self._ctx, n, ks.ctypes.data_as(i32p), _fp(Hp), C.byref(prm),
                                              _fp(W_out) if return_W else None, n_iter.ctypes.data_as(i32p),
                                              viol.ctypes.data_as(dblp),
                                              err.ctypes.data_as(dblp) if prediction_error else None))
        W_list = None
        if return_W:
            offs = np.concatenate([[0], np.cumsum(ks)]).astype(np.int64)
            W_list = [W_out[offs[r] * N:offs[r + 1] * N].reshape(N, ks[r]) for r in range(n)]
        return W_list, n_iter, err

    def nnls_gram(self, H_prod, gram, tol=1e-4, max_iter=1000, alpha_W=0.0, l1_ratio=0.0, n_features=None):
        """Usage refit whose product uses the rows ``H_prod`` (k x G) but whose Gram matrix is GIVEN
        (``gram`` k x k): ``X_sub @ H_sub.T`` for a scaled column subset of the resident matrix is
        ``X @ H_prod.T`` with ``H_prod`` zero outside the subset (cnmf.py:960-975 without a second upload).
        ``n_features`` = the size of that subset: the reference refits on ``tpm[:, hvgs]``, so scikit-learn scales
        ``alpha_W`` by the number of HVGs, not by the width of the resident TPM matrix."""
        if self.shape is None:
            raise RuntimeError("set_matrix() has not been called")
        N, G = self.shape
        Hf = np.ascontiguousarray(self._check_H(H_prod, G), dtype=np.float32)
        k = int(Hf.shape[0])
        g = np.ascontiguousarray(gram, dtype=np.float32)
        if g.shape != (k, k):
            raise ValueError("gram must be (k, k)")
        prm = self._params(tol, max_iter, alpha_W, 0.0, l1_ratio, n_features=n_features)
        W = np.empty((N, k), dtype=np.float32)
        n_iter = C.c_int32(0)
        viol = C.c_double(0.0)
        self._check(self._lib.cnmf_nnls_gram(self._ctx, k, _fp(Hf), _fp(g), C.byref(prm), _fp(W), C.byref(n_iter),
                                             C.byref(viol)))
        return W, int(n_iter.value)

    def nnls_spectra(self, W, tol=1e-4, max_iter=1000, alpha_W=0.0, l1_ratio=0.0):
        """``cNMF.refit_spectra`` (cnmf.py:805-820): NNLS for the spectra (k x G) with the usages ``W`` (N x k)
        fixed, on the resident matrix -- no transposed upload.  ``alpha_W`` plays the role it has in the reference's
        transposed call (it regularises the solved factor, scaled by the N 'features' of X.T)."""
        if self.shape is None:
            raise RuntimeError("set_matrix() has not been called")
        N, G = self.shape
        W = np.ascontiguousarray(W, dtype=np.float64)
        if W.ndim != 2 or W.shape[0] != N:
            raise ValueError("Array with wrong shape passed to NMF (input H). Expected (%d, k), but got %s" % (N, (W.shape,)))
        if not np.isfinite(W).all():
            raise ValueError("Input H contains NaN or infinity.")
        if W.min() < 0:
            raise ValueError("Negative values in data passed to NMF (input H)")
        if W.max() == 0:
            raise ValueError("Array passed to NMF (input H) is full of zeros.")
        k = int(W.shape[1])
        # the transposed problem has G samples and N features: l1_reg_W = N * alpha_W * l1_ratio (sklearn _nmf.py:1254)
        prm = _lib.CdParams(float(tol), int(max_iter), 0, N * alpha_W * l1_ratio, N * alpha_W * (1.0 - l1_ratio),
                            0.0, 0.0, 0, 0)
        H = np.empty((k, G), dtype=np.float64)
        n_iter = C.c_int32(0)
        viol = C.c_double(0.0)
        dblp = C.POINTER(C.c_double)
        self._check(self._lib.cnmf_nnls_spectra(self._ctx, k, W.ctypes.data_as(dblp), C.byref(prm),
                                                H.ctypes.data_as(dblp), C.byref(n_iter), C.byref(viol)))
        return H, int(n_iter.value)

    def nnls_f64(self, H, gram=None, tol=1e-4, max_iter=1000, alpha_W=0.0, l1_ratio=0.0, n_features=None, warn=True):
        """``cnmf_nnls_f64``: the usage refit of :meth:`nnls` with the product ``X @ H.T``, the Gram matrix and the
        coordinate-descent sweeps in FLOAT64 -- what scikit-learn computes when the reference hands it float64 matrices
        (the reference's own tolerance on the consensus tail, sum(diff^2) < 1e-4 on TPM-scale values, needs it).
        ``gram`` (k x k): used instead of ``H @ H.T`` (see :meth:`nnls_gram`).  Returns ``(W [N x k] float64, n_iter)``."""
        if self.shape is None:
            raise RuntimeError("set_matrix() has not been called")
        N, G = self.shape
        Hd = np.ascontiguousarray(self._check_H(H, G), dtype=np.float64)
        k = int(Hd.shape[0])
        dblp = C.POINTER(C.c_double)
        gp = None
        if gram is not None:
            gd = np.ascontiguousarray(gram, dtype=np.float64)
            if gd.shape != (k, k):
                raise ValueError("gram must be (k, k)")
            gp = gd.ctypes.data_as(dblp)
        prm = self._params(tol, max_iter, alpha_W, 0.0, l1_ratio, n_features=n_features)
        W = np.empty((N, k), dtype=np.float64)
        n_iter = C.c_int32(0)
        viol = C.c_double(0.0)
        self._check(self._lib.cnmf_nnls_f64(self._ctx, k, Hd.ctypes.data_as(dblp), gp, C.byref(prm),
                                            W.ctypes.data_as(dblp), C.byref(n_iter), C.byref(viol)))
        if warn and tol > 0 and n_iter.value == max_iter:
            warnings.warn("Maximum number of iterations %d reached. Increase it to improve convergence."
                          % max_iter, ConvergenceWarning)
        return W, int(n_iter.value)

    def xt_matmul_f64(self, W, mean=None, std=None):
        """``W.T @ X`` in float64, or ``W.T @ ((X - mean) / std)`` when ``mean``/``std`` (length G) are given --
        the X^T Y accumulation of ``efficient_ols_all_cols(normalize_y=True)`` (cnmf.py:55-125)."""
        if self.shape is None:
            raise RuntimeError("set_matrix() has not been called")
        N, G = self.shape
        W = np.ascontiguousarray(W, dtype=np.float64)
        if W.ndim != 2 or W.shape[0] != N:
            raise ValueError("W must be (%d, k)" % N)
        k = int(W.shape[1])
        dblp = C.POINTER(C.c_double)
        zs = mean is not None
        out = np.empty((k, G), dtype=np.float64)
        if zs:
            mean = np.ascontiguousarray(mean, dtype=np.float64)
            inv = np.ascontiguousarray(1.0 / np.asarray(std, dtype=np.float64))
            self._check(self._lib.cnmf_xt_matmul_f64(self._ctx, k, W.ctypes.data_as(dblp), 1, mean.ctypes.data_as(dblp),
                                                     inv.ctypes.data_as(dblp), out.ctypes.data_as(dblp)))
        else:
            self._check(self._lib.cnmf_xt_matmul_f64(self._ctx, k, W.ctypes.data_as(dblp), 0, None, None,
                                                     out.ctypes.data_as(dblp)))
        return out

    def kselect_stats(self, spectra_by_k, local_neighborhood_size=0.30, random_state=1, n_init=10, max_iter=300,
                      kmeans_tol=1e-4, nnls_tol=1e-4, nnls_max_iter=1000, alpha_W=0.0, l1_ratio=0.0, store_rows_by_k=None):
        """The statistics loop of ``k_selection_plot`` (cnmf.py:1119-1135) in ONE device call: ``spectra_by_k`` maps
        k -> merged spectra (R_k x G).  Every k goes through the stats branch of the consensus (no density filter,
        KMeans, medians, silhouette), the |K| usage refits run batched (one pass over X), the prediction errors are
        computed with the usages still on the device.  Returns ``{k: dict(silhouette, prediction_error,
        median_spectra, nnls_iter)}``."""
        self._sync_env()
        if self.shape is None:
            raise RuntimeError("set_matrix() has not been called")
        N, G = self.shape
        # ``store_rows_by_k`` (k -> row indices into the resident spectra store) replaces ``spectra_by_k``: nothing is uploaded
        src = store_rows_by_k if store_rows_by_k is not None else spectra_by_k
        ks = sorted(int(k) for k in src)
        n = len(ks)
        if store_rows_by_k is not None:
            if self.spectra_genes != G:
                raise ValueError("the resident store holds spectra over %d genes, the matrix has %d" % (self.spectra_genes, G))
            rows = [np.ascontiguousarray(store_rows_by_k[k], dtype=np.int64).ravel() for k in ks]
            R = np.ascontiguousarray([r.size for r in rows], dtype=np.int32)
            rows_all = np.ascontiguousarray(np.concatenate(rows))
        else:
            S = [np.ascontiguousarray(spectra_by_k[k], dtype=np.float64) for k in ks]
            for k, s in zip(ks, S):
                if s.ndim != 2 or s.shape[1] != G:
                    raise ValueError("spectra for k=%d must be (R, %d)" % (k, G))
            R = np.ascontiguousarray([s.shape[0] for s in S], dtype=np.int32)
            Sp = np.ascontiguousarray(np.concatenate(S, axis=0))
        cprm = (_lib.ConsensusParams * n)()
        us = []
        for i, k in enumerate(ks):
            L = 2 + int(np.log(k))
            us.append(np.random.RandomState(random_state).random_sample(n_init * (1 + (k - 1) * L)))
            cprm[i] = _lib.ConsensusParams(k, int(local_neighborhood_size * int(R[i]) / k), 2.0, 1, 1, int(n_init),
                                           int(max_iter), float(kmeans_tol))
        u = np.ascontiguousarray(np.concatenate(us))
        prm = self._params(nnls_tol, nnls_max_iter, alpha_W, 0.0, l1_ratio)
        ks_a = np.ascontiguousarray(ks, dtype=np.int32)
        sil = np.zeros(n)
        err = np.zeros(n)
        med = np.zeros((int(ks_a.sum()), G))
        nit = np.zeros(n, dtype=np.int32)
        i32p, dblp = C.POINTER(C.c_int32), C.POINTER(C.c_double)
        outs = (cprm, u.ctypes.data_as(dblp), C.byref(prm), sil.ctypes.data_as(dblp), err.ctypes.data_as(dblp),
                med.ctypes.data_as(dblp), nit.ctypes.data_as(i32p))
        if store_rows_by_k is not None:
            rc = self._lib.cnmf_kselect_stats_store(self._ctx, n, ks_a.ctypes.data_as(i32p), R.ctypes.data_as(i32p),
                                                    rows_all.ctypes.data_as(C.POINTER(C.c_int64)), *outs)
        else:
            rc = self._lib.cnmf_kselect_stats(self._ctx, n, ks_a.ctypes.data_as(i32p), R.ctypes.data_as(i32p),
                                              Sp.ctypes.data_as(dblp), *outs)
        self._check(rc)
        out, off = {}, 0
        for i, k in enumerate(ks):
            out[k] = dict(silhouette=float(sil[i]), prediction_error=float(err[i]), median_spectra=med[off:off + k],
                          nnls_iter=int(nit[i]))
            off += k
        return out

    # ------------------------------------------------------------------ NNDSVD init
    def x_matmul(self, Q, trans=False):
        """``X @ Q`` (trans=False, Q is G x c) or ``X.T @ Q`` (trans=True, Q is N x c) on the device."""
        N, G = self.shape
        Q = np.ascontiguousarray(Q, dtype=np.float32)
        if Q.ndim != 2 or Q.shape[0] != (N if trans else G):
            raise ValueError("shape mismatch in x_matmul")
        out = np.empty(((G if trans else N), Q.shape[1]), dtype=np.float32)
        self._check(self._lib.cnmf_x_matmul(self._ctx, int(bool(trans)), _fp(Q), Q.shape[1], _fp(out)))
        return out

    def range_finder(self, Q0_blocks, n_iter, transpose):
        """``cnmf_range_finder``: the power iterations of sklearn's ``randomized_range_finder`` for a GROUP of Gaussian
        start blocks (list of [M_cols, c_b] arrays, sum c_b <= 256) entirely on the device; returns the lists
        ``Q_b`` ([M_rows, c_b], orthonormal columns) and ``B_b = Q_b.T @ M`` ([c_b, M_cols]) in float32."""
        N, G = self.shape
        M_rows, M_cols = (G, N) if transpose else (N, G)
        widths = np.ascontiguousarray([q.shape[1] for q in Q0_blocks], dtype=np.int32)
        Ctot = int(widths.sum())
        Q0 = np.ascontiguousarray(np.concatenate(Q0_blocks, axis=1), dtype=np.float32)
        if Q0.shape != (M_cols, Ctot):
            raise ValueError("start blocks must be [%d, c]" % M_cols)
        Q = np.empty((M_rows, Ctot), dtype=np.float32)
        B = np.empty((Ctot, M_cols), dtype=np.float32)
        self._check(self._lib.cnmf_range_finder(self._ctx, int(bool(transpose)), len(Q0_blocks),
                                                widths.ctypes.data_as(C.POINTER(C.c_int32)), _fp(Q0), int(n_iter),
                                                _fp(Q), _fp(B)))
        offs = np.concatenate([[0], np.cumsum(widths)])
        return ([Q[:, offs[i]:offs[i + 1]] for i in range(len(widths))],
                [B[offs[i]:offs[i + 1]] for i in range(len(widths))])

    def nndsvd_init(self, n_components, random_state=None, eps=1e-6):
        """sklearn's ``init='nndsvd'`` for ONE restart (decomposition/_nmf.py:316-354); see :meth:`nndsvd_init_batch`.
        Returns (W0, H0) in float64 (cast to X's dtype by the caller, like sklearn)."""
        return self.nndsvd_init_batch([n_components], [random_state], eps=eps)[0]

    def nndsvd_init_batch(self, ks, random_states, eps=1e-6, max_cols=256, threads=None, device_range_finder=True):
        """sklearn's ``init='nndsvd'`` (decomposition/_nmf.py:316-354) for MANY restarts: ``_randomized_svd``
        (utils/extmath.py:531-602: n_oversamples=10, n_iter 7|4, LU-normalised power iterations, final QR, small
        SVD, svd_flip), then the positive/negative split.

        The range finders of a whole GROUP of restarts ride in one pass over X: their (k + 10)-column blocks sit side by
        side (up to ``max_cols`` = 256 columns: 13 restarts of rank 9), so the 2 * n_iter + 2 passes over X are paid
        once per group, not once per restart, and (``device_range_finder=True``, the default; needs k + 10 <= 128) the
        normalisation between two passes stays on the device too (``cnmf_range_finder``: Cholesky-QR instead of
        scikit-learn's pivoted LU -- any normaliser leaves the iterated subspace unchanged).  What remains on the host
        per restart is the (k + 10) x G SVD, the sign flip and the positive / negative split.
        ``device_range_finder=False``: every product on the device, LU / QR on a host thread pool (the round-2 scheme,
        batched).  Returns a list of (W0, H0) in float64."""
        from concurrent.futures import ThreadPoolExecutor
        from scipy import linalg
        import os
        N, G = self.shape
        ks = [int(k) for k in ks]
        if len(ks) != len(random_states):
            raise ValueError("need one random_state per restart")
        for k in ks:
            if k > min(N, G):
                raise ValueError("init = 'nndsvd' can only be used when n_components <= min(n_samples, n_features)")
        transpose = N < G                                     # M = X.T when n_samples < n_features
        M_rows, M_cols = (G, N) if transpose else (N, G)
        n_rand = [k + 10 for k in ks]
        n_iters = [7 if k < 0.1 * min(N, G) else 4 for k in ks]
        if max(n_rand) > max_cols:
            raise NotImplementedError("n_components + 10 = %d columns exceed one device pass (%d)" % (max(n_rand), max_cols))
        if threads is None:                                  # CPUs this process may really use (affinity, cgroup quota)
            try:
                threads = len(os.sched_getaffinity(0))
            except AttributeError:
                threads = os.cpu_count() or 1
            try:
                quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
                if quota != "max":
                    threads = max(1, min(threads, int(float(quota) / float(period))))
            except Exception:
                pass
        pool = ThreadPoolExecutor(max(1, min(threads, 64)))
        # one LAPACK thread per factorization: the parallelism is ACROSS the restarts of a group (a BLAS that spawns its
        # own 64 threads inside each of 16 pool threads runs slower than one thread would)
        try:
            from threadpoolctl import threadpool_limits
            blas_limit = threadpool_limits(1)
        except Exception:
            blas_limit = None

        def passes(Qs, trans):
            """[M_r @ Q_r] (trans=False) or [M_r.T @ Q_r] for the blocks of one group: ONE device product."""
            out = self.x_matmul(np.concatenate(Qs, axis=1), trans=trans).astype(np.float64)
            offs = np.cumsum([0] + [q.shape[1] for q in Qs])
            return [out[:, offs[i]:offs[i + 1]] for i in range(len(Qs))]

        def lu_pl(A):
            return linalg.lu(A, permute_l=True, check_finite=False)[0]

        def finish(args):
            k, Q, B = args
            return _nndsvd_finish(k, Q, B, transpose, eps)

        results = [None] * len(ks)
        # groups of restarts whose blocks fit one pass AND that make the same number of power iterations
        order = sorted(range(len(ks)), key=lambda r: (n_iters[r], ks[r]))
        groups, cur, cols = [], [], 0
        for r in order:
            if cur and (cols + n_rand[r] > max_cols or n_iters[r] != n_iters[cur[0]]):
                groups.append(cur); cur, cols = [], 0
            cur.append(r); cols += n_rand[r]
        if cur:
            groups.append(cur)
        on_device = device_range_finder and max(n_rand) <= _lib.CNMF_KMAX
        pending = []

        def collect(item):
            for r, fut in zip(*item):
                results[r] = fut.result()

        try:
            for grp in groups:
                Qs = []
                for r in grp:
                    rs = random_states[r]
                    rng = rs if isinstance(rs, np.random.RandomState) else np.random.RandomState(rs)
                    Qs.append(rng.normal(size=(M_cols, n_rand[r])))
                if on_device:
                    Qd, Bd = self.range_finder(Qs, n_iters[grp[0]], transpose)
                    # the host tails of this group run on the pool WHILE the device finds the ranges of the next group
                    # (the ctypes call releases the GIL); at most two groups are in flight
                    pending.append((grp, [pool.submit(finish, (ks[r], Q, B)) for r, Q, B in zip(grp, Qd, Bd)]))
                    if len(pending) > 2:
                        collect(pending.pop(0))
                    continue
                for _ in range(n_iters[grp[0]]):
                    Qs = list(pool.map(lu_pl, passes(Qs, trans=transpose)))            # Q = PL of M @ Q
                    Qs = list(pool.map(lu_pl, passes(Qs, trans=not transpose)))        # Q = PL of M.T @ Q
                Qs = list(pool.map(lambda A: linalg.qr(A, mode="economic", check_finite=False)[0], passes(Qs, trans=transpose)))
                Bs = [b.T for b in passes(Qs, trans=not transpose)]                    # Q.T @ M
                for r, wh in zip(grp, pool.map(finish, [(ks[r], Q, B) for r, Q, B in zip(grp, Qs, Bs)])):
                    results[r] = wh
            while pending:
                collect(pending.pop(0))
        finally:
            pool.shutdown(wait=True)
            if blas_limit is not None:
                blas_limit.restore_original_limits()
        return results

    # ------------------------------------------------------------------ consensus core
    def consensus(self, spectra, k, density_threshold=0.5, local_neighborhood_size=0.30,
                  skip_density=False, want_silhouette=False, random_state=1, n_init=10,
                  max_iter=300, tol=1e-4, return_dist=False, store_rows=None):
        """Numerical core of ``cNMF.consensus`` (cnmf.py:871-916) on the device, float64.

        ``spectra``: the merged per-restart spectra (R x G) -- or ``None`` with ``store_rows`` (R row indices into the
        context's resident spectra store, see ``nmf_batch(resident="keep")``): the merged spectra are then gathered on
        the device, nothing is uploaded.  Returns a dict with
        ``local_density`` (R,), ``density_filter`` (R,) bool, ``labels`` (R,) int (0-based,
        -1 = filtered), ``median_spectra`` (k x G, rows sum to 1), ``inertia``,
        ``silhouette`` (if requested), ``topics_dist`` (R x R, if requested)."""
        self._sync_env()
        if store_rows is not None:
            rows = np.ascontiguousarray(store_rows, dtype=np.int64).ravel()
            R, G = int(rows.size), self.spectra_genes
        else:
            S = np.ascontiguousarray(spectra, dtype=np.float64)
            if S.ndim != 2:
                raise ValueError("spectra must be 2-D")
            R, G = S.shape
        k = int(k)
        n_neighbors = int(local_neighborhood_size * R / k)                 # cnmf.py:879
        L = 2 + int(np.log(k))
        # the draws KMeans(random_state=1) takes: data-independent count, so the whole
        # stream is generated up front with numpy's legacy RNG (bit-identical to sklearn)
        u = np.random.RandomState(random_state).random_sample(n_init * (1 + (k - 1) * L))
        prm = _lib.ConsensusParams(k, n_neighbors, float(density_threshold), int(bool(skip_density)),
                                   int(bool(want_silhouette)), int(n_init), int(max_iter), float(tol))
        dens = np.zeros(R, dtype=np.float64)
        keep = np.zeros(R, dtype=np.int32)
        labels = np.zeros(R, dtype=np.int32)
        med = np.zeros((k, G), dtype=np.float64)
        dist = np.zeros((R, R), dtype=np.float64) if return_dist else None
        stats = np.zeros(4, dtype=np.float64)
        dblp, i32p = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        tail_args = (R, G, C.byref(prm), u.ctypes.data_as(dblp), dens.ctypes.data_as(dblp),
                     keep.ctypes.data_as(i32p), labels.ctypes.data_as(i32p), med.ctypes.data_as(dblp),
                     dist.ctypes.data_as(dblp) if return_dist else None, stats.ctypes.data_as(dblp))
        if store_rows is not None:
            rc = self._lib.cnmf_consensus_store(self._ctx, rows.ctypes.data_as(C.POINTER(C.c_int64)), *tail_args)
        else:
            rc = self._lib.cnmf_consensus(self._ctx, S.ctypes.data_as(dblp), *tail_args)
        if rc == -4 and b"Zero components remain" in self._lib.cnmf_last_error(self._ctx):
            raise RuntimeError("Zero components remain after density filtering. Consider increasing density threshold")
        self._check(rc)
        out = dict(local_density=dens, density_filter=keep.astype(bool), labels=labels,
                   median_spectra=med, inertia=float(stats[1]), n_kept=int(stats[0]),
                   kmeans_n_iter=int(stats[3]), n_neighbors=n_neighbors)
        if want_silhouette:
            out["silhouette"] = float(stats[2])
        if return_dist:
            out["topics_dist"] = dist
        return out

    def pairwise_distances(self, rows, labels=None, return_dist=True):
        """``cnmf_pairwise_distances``: ``sklearn.metrics.euclidean_distances(rows)`` (cnmf.py:891, 988) and / or
        ``silhouette_score(rows, labels, metric='euclidean')`` (cnmf.py:923) on the device in float64, rows as given.
        Returns ``(D or None, silhouette or None)``."""
        self._sync_env()
        rows = np.ascontiguousarray(rows, dtype=np.float64)
        if rows.ndim != 2:
            raise ValueError("rows must be 2-D")
        R, G = rows.shape
        dblp = C.POINTER(C.c_double)
        D = np.empty((R, R), dtype=np.float64) if return_dist else None
        sil, lp, k = None, None, 0
        if labels is not None:
            lab = np.asarray(labels)
            if lab.shape != (R,):
                raise ValueError("labels must have one entry per row")
            uniq, inv = np.unique(lab, return_inverse=True)          # LabelEncoder, sklearn metrics/cluster/_unsupervised.py
            k = len(uniq)
            if not 1 < k < R:
                raise ValueError("Number of labels is %d. Valid values are 2 to n_samples - 1 (inclusive)" % k)
            inv = np.ascontiguousarray(inv, dtype=np.int32)
            lp = inv.ctypes.data_as(C.POINTER(C.c_int32))
            sil = C.c_double(0.0)
        self._check(self._lib.cnmf_pairwise_distances(self._ctx, rows.ctypes.data_as(dblp), R, G, lp, k,
                                                      D.ctypes.data_as(dblp) if return_dist else None,
                                                      C.byref(sil) if sil is not None else None))
        return D, (float(sil.value) if sil is not None else None)

    def prediction_error(self, W, H):
        """``((X - W @ H)**2).sum()`` over the resident matrix (cnmf.py:926-930)."""
        if self.shape is None:
            raise RuntimeError("set_matrix() has not been called")
        N, G = self.shape
        W = np.ascontiguousarray(W, dtype=np.float64)
        H = np.ascontiguousarray(H, dtype=np.float64)
        if W.shape[0] != N or H.shape[1] != G or W.shape[1] != H.shape[0]:
            raise ValueError("shape mismatch: W %s, H %s, X %s" % (W.shape, H.shape, (N, G)))
        err = C.c_double(0.0)
        dblp = C.POINTER(C.c_double)
        self._check(self._lib.cnmf_prediction_error(self._ctx, int(H.shape[0]), W.ctypes.data_as(dblp),
                                                    H.ctypes.data_as(dblp), C.byref(err)))
        return float(err.value)

    # ------------------------------------------------------------------ diagnostics
    # ------------------------------------------------------------------ multi-GPU exchange (RCCL)
    def comm_unique_id(self):
        """128-byte RCCL id (rank 0 creates it; ship it to the other ranks out of band)."""
        buf = (C.c_ubyte * _lib.COMM_ID_BYTES)()
        self._check(self._lib.cnmf_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, unique_id, rank, world):
        """Collective: every rank calls it with the same id (ncclCommInitRank)."""
        if len(unique_id) != _lib.COMM_ID_BYTES:
            raise ValueError("unique_id must be %d bytes" % _lib.COMM_ID_BYTES)
        buf = (C.c_ubyte * _lib.COMM_ID_BYTES).from_buffer_copy(unique_id)
        self._check(self._lib.cnmf_comm_init(self._ctx, buf, int(rank), int(world)))

    def comm_finalize(self):
        self._check(self._lib.cnmf_comm_finalize(self._ctx))

    @property
    def comm_rank(self):
        return int(self._lib.cnmf_comm_rank(self._ctx))

    @property
    def comm_world(self):
        return int(self._lib.cnmf_comm_world(self._ctx))

    def allgather_array(self, a):
        """All-gather equally-shaped host arrays: returns ``[world, *a.shape]``."""
        a = np.ascontiguousarray(a)
        out = np.empty((self.comm_world,) + a.shape, dtype=a.dtype)
        self._check(self._lib.cnmf_allgather_bytes(self._ctx, a.ctypes.data_as(C.c_void_p), a.nbytes,
                                                   out.ctypes.data_as(C.c_void_p)))
        return out

    def allgather_spectra(self, local, rows_max, n_genes=None):
        """THE data-path collective: one all-gather of this rank's packed float32 spectra
        (``local`` [rows, G], or ``None`` for the context's resident store) zero-padded to
        ``rows_max`` rows.  Returns ``[world, rows_max, G]`` float32."""
        if local is None:
            rows, G = self.spectra_rows, (self.spectra_genes or self.shape[1])
            lp = None
        else:
            local = np.ascontiguousarray(local, dtype=np.float32)
            rows, G = (int(local.shape[0]), int(local.shape[1] if local.ndim == 2 else n_genes))
            lp = _fp(local) if rows else None
            if not rows:
                lp = _fp(np.zeros(1, np.float32))
        if n_genes is not None and int(n_genes) != G:
            raise ValueError("n_genes does not match the block")
        out = np.empty((self.comm_world, int(rows_max), G), dtype=np.float32)
        if rows_max:
            self._check(self._lib.cnmf_allgather_spectra(self._ctx, lp, rows, int(rows_max), G, _fp(out)))
        return out

    @property
    def spectra_rows(self):
        """Rows in the resident spectra store (``nmf_batch(..., resident=True)`` appends to it)."""
        return int(self._lib.cnmf_spectra_rows(self._ctx))

    def spectra_reset(self):
        self._check(self._lib.cnmf_spectra_reset(self._ctx))
        self.store_gen += 1                      # row indices handed out before this point are void

    @property
    def spectra_genes(self):
        return int(self._lib.cnmf_spectra_genes(self._ctx))

    def spectra_append(self, rows):
        """Upload ``rows`` (n x G, float32) behind the store's content; returns the index of the first new row."""
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        if rows.ndim != 2:
            raise ValueError("rows must be 2-D")
        first = self.spectra_rows
        self._check(self._lib.cnmf_spectra_append(self._ctx, _fp(rows), rows.shape[0], rows.shape[1]))
        return first

    def spectra_fetch(self, row0=0, rows=None):
        """Rows ``[row0, row0 + rows)`` of the resident spectra store on the host (default: all of it)."""
        total = self.spectra_rows
        rows = total - int(row0) if rows is None else int(rows)
        out = np.empty((rows, self.spectra_genes if total else 0), dtype=np.float32)
        if out.size:
            self._check(self._lib.cnmf_spectra_fetch_rows(self._ctx, int(row0), rows, _fp(out)))
        return out

    def debug_gemm(self, mode, A, B, variant=0, nsplit=1, reps=0):
        A = np.ascontiguousarray(A, dtype=np.float32)
        B = np.ascontiguousarray(B, dtype=np.float32)
        KC, K = A.shape
        J = B.shape[0] if mode == 0 else B.shape[1]
        Cout = np.empty((KC, J), dtype=np.float32)
        ms = C.c_double(0.0)
        self._check(self._lib.cnmf_debug_gemm(self._ctx, mode, variant, _fp(A), _fp(B), _fp(Cout),
                                              KC, K, J, nsplit, C.byref(ms), reps))
        return Cout, ms.value

    def debug_gemm3(self, A, B, nsplit=1, reps=0):
        """A [KC, K] . B [J, K]^T through the split-operand bf16 MFMA path; returns (C, ms)."""
        A = np.ascontiguousarray(A, dtype=np.float32)
        B = np.ascontiguousarray(B, dtype=np.float32)
        KC, K = A.shape
        J = B.shape[0]
        Cout = np.empty((KC, J), dtype=np.float32)
        ms = C.c_double(0.0)
        self._check(self._lib.cnmf_debug_gemm3(self._ctx, _fp(A), _fp(B), _fp(Cout), KC, K, J, int(nsplit),
                                               C.byref(ms), int(reps)))
        return Cout, ms.value

    def debug_gemm3c(self, A, Bn, nsplit=1, reps=0):
        """A [KC, K] . Bn [J, K]^T through the count-path kernel (Bn: integers <= 256); returns (C, ms)."""
        A = np.ascontiguousarray(A, dtype=np.float32)
        Bn = np.ascontiguousarray(Bn, dtype=np.float32)
        KC, K = A.shape
        J = Bn.shape[0]
        Cout = np.empty((KC, J), dtype=np.float32)
        ms = C.c_double(0.0)
        self._check(self._lib.cnmf_debug_gemm3c(self._ctx, _fp(A), _fp(Bn), _fp(Cout), KC, K, J, int(nsplit),
                                                C.byref(ms), int(reps)))
        return Cout, ms.value

    def debug_gemm2h(self, A, Bn, nsplit=1, nsub=2, reps=0, sweep_bound=False):
        """A [KC, K] . Bn [J, K]^T through the f16 two-plane count kernel (A >= 0, Bn: integers <= 65535);
        returns (C, ms).  ``sweep_bound=True`` scales the rows of A by the bound the W half-step reports
        (sqrt(sum w^2) per 1024-entry block) instead of the exact row maximum -- the production pass-B scaling."""
        nsub = int(nsub) | (128 if sweep_bound else 0)
        A = np.ascontiguousarray(A, dtype=np.float32)
        Bn = np.ascontiguousarray(Bn, dtype=np.float32)
        KC, K = A.shape
        J = Bn.shape[0]
        Cout = np.empty((KC, J), dtype=np.float32)
        ms = C.c_double(0.0)
        self._check(self._lib.cnmf_debug_gemm2h(self._ctx, _fp(A), _fp(Bn), _fp(Cout), KC, K, J, int(nsplit),
                                                int(nsub), C.byref(ms), int(reps)))
        return Cout, ms.value

    def debug_standard_normal(self, seed, n):
        out = np.empty(max(n, 1), dtype=np.float64)
        self._check(self._lib.cnmf_debug_standard_normal(self._ctx, C.c_uint32(seed), n,
                                                         out.ctypes.data_as(C.POINTER(C.c_double))))
        return out[:n]
