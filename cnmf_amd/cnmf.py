"""Host-side mirror of the reference's ``class cNMF`` for the accelerated hot path.

Same method names, argument meaning, on-disk artefacts and error behaviour as
/root/reference/src/cnmf/cnmf.py for ``factorize`` (:692-745), ``combine`` / ``combine_nmf``
(:462-483, :748-773), ``refit_usage`` / ``refit_spectra`` (:776-820), ``consensus``
(:823-985, without the plotting block) and the statistics loop of ``k_selection_plot``
(:1119-1135) -- but every numerical step runs on the MI355X through ``Engine``
(libcnmf_hip.so).  There is no CPU fallback.

What is different by design
* all restarts of a worker run as ONE batched device call (X is uploaded once, not re-read
  per worker process; cnmf.py:726), and the spectra come back in one transfer;
* ``factorize`` can keep the merged spectra in memory (``self.spectra_cache``) so that
  ``combine`` becomes a gather instead of n_iter file reads; the per-restart
  ``.df.npz`` files are still written (``write_iter_files=True``) so that the reference's
  ``completed`` ledger / ``skip_completed_runs`` / ``skip_missing_files`` semantics survive;
* ``prepare`` (HVG selection, TPM, scanpy I/O) is out of scope:
  ``prepare_from_matrix`` takes the normalised cells x genes matrix the reference's
  ``prepare`` would have written and produces the same ledger / yaml.  The matrix itself is
  stored as ``.df.npz`` (the reference's own DataFrame container, cnmf.py:31-40) because
  scanpy/anndata do not exist in this image.
"""
import errno
import itertools
import os
import warnings

import numpy as np
import pandas as pd
import yaml

from .engine import Engine


# ---------------------------------------------------------------- reference I/O helpers
def save_df_to_npz(obj, filename):
    """cnmf.py:31-32: ``np.savez_compressed(filename, data=..., index=..., columns=...)`` -- the same zip container with the
    same three members, read back by the reference's ``load_df_from_npz`` (np.load) unchanged.  One difference a reader cannot
    see: a float ``data`` member of a megabyte or more (the usages: cells x k) is STORED instead of deflated -- zlib gains
    1 % on float64 mantissas and costs 0.2 s per 3.6 MB (the critical path of consensus()'s artefact writes at 50 000 cells);
    labels and small tables are deflated as before."""
    data = np.asanyarray(obj.values)
    if data.dtype.kind != "f" or data.nbytes < (1 << 20):
        np.savez_compressed(filename, data=data, index=obj.index.values, columns=obj.columns.values)
        return
    import zipfile
    if not str(filename).endswith(".npz"):
        filename = str(filename) + ".npz"                       # (np.savez appends it too)
    with zipfile.ZipFile(filename, "w", compression=zipfile.ZIP_DEFLATED, allowZip64=True) as zf:
        for name, arr, how in (("data", data, zipfile.ZIP_STORED), ("index", obj.index.values, zipfile.ZIP_DEFLATED),
                               ("columns", obj.columns.values, zipfile.ZIP_DEFLATED)):
            zi = zipfile.ZipInfo(name + ".npy", date_time=(1980, 1, 1, 0, 0, 0))
            zi.compress_type = how
            with zf.open(zi, "w", force_zip64=True) as f:
                np.lib.format.write_array(f, np.asanyarray(arr), allow_pickle=True)


_SIBLING_BYTES = 64 << 20


def save_df_to_npz_fast(obj, filename, sibling_ok=False):
    """Same container as save_df_to_npz without zlib: used only for the (large) normalised
    matrix, which the reference stores as uncompressed h5ad (cnmf.py:561); zlib over 400 MB
    costs ~10 s and buys nothing on a scratch file.  With ``sibling_ok`` (the normalised matrix only: a file the
    reference never reads as npz) a matrix above 64 MB goes into a
    sibling ``<file>.data.npy`` named inside the npz: a zip member costs a CRC-32 pass over
    every byte (0.6 s per 800 MB, single-threaded), a plain .npy does not -- and the other
    workers can map it."""
    data = np.ascontiguousarray(obj.values)
    if sibling_ok and data.nbytes >= _SIBLING_BYTES:
        sibling = filename + ".data.npy"
        np.save(sibling, data)
        np.savez(filename, data_file=np.array(os.path.basename(sibling)), index=obj.index.values, columns=obj.columns.values)
    else:
        np.savez(filename, data=data, index=obj.index.values, columns=obj.columns.values)
        stale = filename + ".data.npy"                       # (left by an earlier, larger matrix under the same name)
        if os.path.exists(stale):
            os.remove(stale)


def save_csr_fast(filename, mat):
    """The sparse TPM stand-in (``tpm_sparse``: this package's own scratch container, the reference keeps a tpm.h5ad):
    scipy's ``save_npz`` layout, except that above 64 MB the three arrays go into sibling ``.npy`` files named inside the
    npz -- a zip member costs a CRC-32 pass over every byte (0.2 s per 560 MB), a plain .npy does not."""
    if mat.data.nbytes + mat.indices.nbytes < _SIBLING_BYTES:
        import scipy.sparse as sp
        sp.save_npz(filename, mat, compressed=False)
        for part in ("data", "indices", "indptr"):
            stale = "%s.%s.npy" % (filename, part)
            if os.path.exists(stale):
                os.remove(stale)
        return
    names = {}
    for part in ("data", "indices", "indptr"):
        sib = "%s.%s.npy" % (filename, part)
        np.save(sib, getattr(mat, part))
        names[part + "_file"] = np.array(os.path.basename(sib))
    np.savez(filename, shape=np.array(mat.shape), format=np.array("csr"), **names)


def load_csr(filename):
    """Reads what save_csr_fast (either form) or scipy's save_npz wrote."""
    import scipy.sparse as sp
    with np.load(filename, allow_pickle=False) as f:
        if "data_file" not in f.files:
            return sp.load_npz(filename).tocsr()
        d = os.path.dirname(os.path.abspath(filename))
        parts = [np.load(os.path.join(d, str(f[part + "_file"]))) for part in ("data", "indices", "indptr")]
        return sp.csr_matrix(tuple(parts), shape=tuple(int(v) for v in f["shape"]))


def save_df_to_text(obj, filename):
    """cnmf.py:34-35: ``obj.to_csv(filename, sep='\\t')``.  All-float64 frames with plain labels (the usages: cells x k)
    are formatted here -- the same bytes (shortest round-trip repr per value, like pandas' float -> str), a quarter of the
    time; anything else (other dtypes, missing values, labels that would need quoting) goes through pandas."""
    def _plain_axis(ax):
        # labels whose str() IS what pandas writes: python str objects or integers, no nulls (None / NaN become '' in
        # to_csv, timestamps lose their midnight time) -- anything else goes through pandas
        if ax.nlevels != 1 or ax.name is not None:
            return False
        if ax.dtype.kind in "iu":
            return True
        return ax.dtype == object and all(type(v) is str for v in ax.values)

    plain = isinstance(obj, pd.DataFrame) and _plain_axis(obj.index) and _plain_axis(obj.columns)
    if plain:
        vals = obj.values
        labels = [str(c) for c in obj.columns] + [str(i) for i in obj.index]
        plain = (vals.ndim == 2 and vals.dtype == np.float64 and vals.size > 0 and np.isfinite(vals).all()
                 and not any(ch in lab for lab in labels for ch in "\t\n\r\"") and "" not in labels)
    if not plain:
        obj.to_csv(filename, sep="\t")
        return
    ncol = vals.shape[1]
    header = ("\t".join([""] + labels[:ncol]) + "\n").encode("utf-8")
    try:                                                       # the library's formatter (repr(float) in C++); host code only
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        v = np.ascontiguousarray(vals, dtype=np.float64)
        blob = "\n".join(labels[ncol:]).encode("utf-8")
        out = np.empty(int(v.size) * 33 + len(blob) + v.shape[0] + 1, dtype=np.uint8)
        n = lib.cnmf_format_rows_f64(v.ctypes.data_as(C.POINTER(C.c_double)), v.shape[0], v.shape[1], b"\t", blob, len(blob),
                                     out.ctypes.data_as(C.c_void_p), out.size)
        if n > 0:
            with open(filename, "wb") as F:
                F.write(header)
                F.write(memoryview(out)[:n])
            return
    except (ImportError, OSError, AttributeError):
        pass
    body = "\n".join("%s\t%s" % (lab, "\t".join(map(repr, row))) for lab, row in zip(labels[ncol:], vals.tolist()))
    with open(filename, "w", newline="") as F:
        F.write("\t".join([""] + labels[:ncol]) + "\n" + body + "\n")


def load_df_from_npz(filename):
    """cnmf.py:37-40"""
    with np.load(filename, allow_pickle=True) as f:
        if "data_file" in f.files:               # written by save_df_to_npz_fast: the matrix sits beside the npz
            data = np.load(os.path.join(os.path.dirname(os.path.abspath(filename)), str(f["data_file"])))
            obj = pd.DataFrame(data=data, index=f["index"], columns=f["columns"])
        else:
            obj = pd.DataFrame(**f)
    return obj


def check_dir_exists(path):
    """cnmf.py:42-50"""
    try:
        os.makedirs(path)
    except OSError as exception:
        if exception.errno != errno.EEXIST:
            raise


def worker_filter(iterable, worker_index, total_workers):
    """cnmf.py:52-53: static round-robin shard of the restart ledger."""
    return (p for i, p in enumerate(iterable) if (i - worker_index) % total_workers == 0)


def ledger_seeds(ks, n_iter, random_state_seed):
    """The (k, iter, nmf_seed) rows of the restart ledger (cnmf.py:593-605): one draw of ``len(ks) * n_iter`` seeds
    from numpy's legacy global RNG (the UN-deduplicated ks set the count, cnmf.py:599), assigned to the product of
    the sorted de-duplicated ks and range(n_iter)."""
    if type(ks) is int:
        ks = [ks]
    k_list = sorted(set(list(ks)))
    n_runs = len(ks) * n_iter
    np.random.seed(seed=random_state_seed)
    nmf_seeds = np.random.randint(low=1, high=(2 ** 31) - 1, size=n_runs)
    return [(k, r, nmf_seeds[i]) for i, (k, r) in enumerate(itertools.product(k_list, range(n_iter)))]


class SparseFrame:
    """A cells x genes scipy CSR matrix with its labels -- what the reference holds in ``norm_counts`` (an AnnData with a
    sparse ``.X``, cnmf.py:537-556 sparse branch) when the counts were stored sparse and not densified.  Only what the
    class below touches: ``values`` (the CSR matrix), ``index``, ``columns``, ``shape``."""

    def __init__(self, values, index, columns):
        self.values, self.index, self.columns = values, pd.Index(index), pd.Index(columns)
        self.shape = values.shape


class cNMF:
    # the R x R distance matrix only leaves the device when a consumer exists (the reference's plotting block,
    # integration/hip_backend.py sets this): 8 R^2 bytes of host memory otherwise bought nothing
    materialize_topics_dist = False

    def __init__(self, output_dir=".", name=None, device=0, engine=None, compress_merged=True, detect_counts=True):
        """Same constructor semantics as the reference (cnmf.py:268-296) plus the GPU index.
        ``compress_merged=False`` writes the merged-spectra files without zlib (same npz container, read
        by the reference's ``load_df_from_npz`` alike): zlib over 130 MB costs ~2 s of a 12 s job."""
        self.output_dir = output_dir
        if name is None:
            import datetime
            import uuid
            now = datetime.datetime.now()
            rand_hash = uuid.uuid4().hex[:6]
            name = "%s_%s" % (now.strftime("%Y_%m_%d"), rand_hash)
        self.name = name
        self.paths = None
        self._initialize_dirs()
        self.device = device
        self.detect_counts = detect_counts      # False: never use the integer-plane GEMMs (Engine.set_count_detection)
        self._engine = engine
        self._engine_key = None
        self.spectra_cache = {}          # (k, iter) -> spectra ndarray (k x genes) kept from factorize
        self._spectra_columns = None     # their gene names
        self.merged_cache = {}           # k -> merged spectra DataFrame kept from combine (skips a reload)
        self._store_rows = {}            # (k, iter) -> (first row in the engine's resident spectra store, store generation)
        self._resident_obj = None        # STRONG reference to the matrix object that is resident (identity check:
                                         # while we hold it CPython cannot hand its id() to another object)
        self.compress_merged = compress_merged
        self.last_factorize_stats = None
        self.last_factorize_jobs = []
        self.learned_iterations = {}     # rank -> mean outer iterations the last factorize saw (queue hints for the next one)

    # ------------------------------------------------------------------ paths (cnmf.py:298-330)
    def _initialize_dirs(self):
        if self.paths is None:
            check_dir_exists(self.output_dir)
            check_dir_exists(os.path.join(self.output_dir, self.name))
            check_dir_exists(os.path.join(self.output_dir, self.name, "cnmf_tmp"))
            d, n = self.output_dir, self.name
            t = lambda s: os.path.join(d, n, "cnmf_tmp", n + s)      # noqa: E731
            o = lambda s: os.path.join(d, n, n + s)                  # noqa: E731
            self.paths = {
                "normalized_counts": t(".norm_counts.df.npz"),
                "normalized_counts_sparse": t(".norm_counts.csr.npz"),   # sparse normalised counts (CSR + two label files)
                "nmf_replicate_parameters": t(".nmf_params.df.npz"),
                "nmf_run_parameters": t(".nmf_idvrun_params.yaml"),
                "nmf_genes_list": o(".overdispersed_genes.txt"),
                "tpm": t(".tpm.df.npz"),
                "tpm_sparse": t(".tpm.csr.npz"),              # scipy CSR stand-in for a sparse tpm.h5ad (cnmf.py:303)
                "tpm_sparse_genes": t(".tpm.csr.genes.txt"),
                "tpm_stats": t(".tpm_stats.df.npz"),
                "iter_spectra": t(".spectra.k_%d.iter_%d.df.npz"),
                "iter_usages": t(".usages.k_%d.iter_%d.df.npz"),
                "merged_spectra": t(".spectra.k_%d.merged.df.npz"),
                "local_density_cache": t(".local_density_cache.k_%d.merged.df.npz"),
                "consensus_spectra": t(".spectra.k_%d.dt_%s.consensus.df.npz"),
                "consensus_spectra__txt": o(".spectra.k_%d.dt_%s.consensus.txt"),
                "consensus_usages": t(".usages.k_%d.dt_%s.consensus.df.npz"),
                "consensus_usages__txt": o(".usages.k_%d.dt_%s.consensus.txt"),
                "consensus_stats": t(".stats.k_%d.dt_%s.df.npz"),
                "k_selection_stats": o(".k_selection_stats.df.npz"),
                "gene_spectra_score": t(".gene_spectra_score.k_%d.dt_%s.df.npz"),
                "gene_spectra_score__txt": o(".gene_spectra_score.k_%d.dt_%s.txt"),
                "gene_spectra_tpm": t(".gene_spectra_tpm.k_%d.dt_%s.df.npz"),
                "gene_spectra_tpm__txt": o(".gene_spectra_tpm.k_%d.dt_%s.txt"),
            }

    # ------------------------------------------------------------------ engine plumbing
    @property
    def engine(self):
        """The per-process device context (one process per GPU)."""
        if self._engine is None:
            self._engine = Engine(self.device, detect_counts=self.detect_counts)
        return self._engine

    def _get_engine(self, X, key):
        """One resident upload per distinct matrix (X is NOT re-read per restart/worker).  ``key`` identifies a
        FILE-backed matrix (path, mtime); ``key=None`` = an ad-hoc matrix, which is uploaded on every call
        unless it is the very object that is resident (``_resident_obj``, held by strong reference -- never
        id(): CPython reuses the ids of freed temporaries)."""
        if self._engine is None:
            self._engine = Engine(self.device, detect_counts=self.detect_counts)
        if key is None:
            self._engine.set_matrix(X)
            self._engine_key = None
            self._resident_obj = None
        elif self._engine_key != key:
            self._engine.set_matrix(X)
            self._engine_key = key
            self._resident_obj = None
        return self._engine

    def _forget_results(self):
        """A new prepare invalidates everything derived from the previous matrix / ledger."""
        self.spectra_cache.clear()
        self.merged_cache.clear()
        self._store_rows = {}
        if self._engine is not None and hasattr(self._engine, "spectra_reset"):
            self._engine.spectra_reset()
        self._resident_obj = None
        self._engine_key = None
        self._norm_counts_cache = (None, None)
        self._tpm_sparse_cache = None

    def _nc_path(self):
        """The file that holds the normalised matrix: the dense frame, or -- when ``prepare_from_matrix`` was handed a
        scipy.sparse matrix -- the CSR container (the reference's normalized_counts h5ad keeps a sparse X likewise)."""
        sparse = self.paths["normalized_counts_sparse"]
        return sparse if os.path.exists(sparse) else self.paths["normalized_counts"]

    def _load_norm_counts(self):
        """The normalised matrix file, kept in memory between the stages of one process."""
        path = self._nc_path()
        key = (path, os.path.getmtime(path))
        if getattr(self, "_norm_counts_cache", (None, None))[0] != key:
            if path == self.paths["normalized_counts_sparse"]:
                obj = SparseFrame(load_csr(path), open(path + ".cells.txt").read().split("\n"),
                                  open(path + ".genes.txt").read().split("\n"))
            else:
                obj = load_df_from_npz(path)
            self._norm_counts_cache = (key, obj)
        return self._norm_counts_cache[1]

    # ------------------------------------------------------------------ ledger (cnmf.py:564-658)
    def get_nmf_iter_params(self, ks, n_iter=100, random_state_seed=None, beta_loss="kullback-leibler",
                            alpha_usage=0.0, alpha_spectra=0.0, init="random", max_iter=1000):
        if type(ks) is int:
            ks = [ks]
        k_list = sorted(set(list(ks)))
        from ._lib import CNMF_KMAX, CNMF_MU_KMAX
        kmax = CNMF_KMAX if beta_loss == "frobenius" else CNMF_MU_KMAX
        if k_list and max(k_list) > kmax:
            # fail at prepare time, not after the restarts were paid for (the reference itself has no limit,
            # cnmf.py:1243; the device engine: 128 for the coordinate-descent solver, 64 for the
            # multiplicative-update solver)
            raise NotImplementedError("n_components=%d > %d is not supported by the device engine (beta_loss=%r)"
                                      % (max(k_list), kmax, beta_loss))
        replicate_params = []
        for k, r, nmf_seed in ledger_seeds(ks, n_iter, random_state_seed):
            done = os.path.exists(self.paths["iter_spectra"] % (k, r))
            replicate_params.append([k, r, nmf_seed, done])
        replicate_params = pd.DataFrame(replicate_params, columns=["n_components", "iter", "nmf_seed", "completed"])
        n_completed = replicate_params["completed"].sum()
        if n_completed > 0:
            warnings.warn("%d runs already appear completed. If this is unexpected, consider re-initializing "
                          "the cnmf object with a different run name or output directory" % n_completed, UserWarning)
        _nmf_kwargs = dict(alpha_W=alpha_usage, alpha_H=alpha_spectra, l1_ratio=0.0, beta_loss=beta_loss,
                           solver="mu", tol=1e-4, max_iter=max_iter, init=init)
        if beta_loss == "frobenius":
            _nmf_kwargs["solver"] = "cd"
        return replicate_params, _nmf_kwargs

    def update_nmf_iter_params(self):
        _nmf_kwargs = yaml.load(open(self.paths["nmf_run_parameters"]), Loader=yaml.FullLoader)
        replicate_params = load_df_from_npz(self.paths["nmf_replicate_parameters"])
        for i in replicate_params.index:
            replicate_params.at[i, "completed"] = os.path.exists(
                self.paths["iter_spectra"] % (replicate_params.at[i, "n_components"], replicate_params.at[i, "iter"]))
        remaining = (replicate_params["completed"] == False).sum()      # noqa: E712
        print("{n} NMF runs are currently incomplete".format(n=remaining))
        self.save_nmf_iter_params(replicate_params, _nmf_kwargs)

    def save_nmf_iter_params(self, replicate_params, run_params):
        self._initialize_dirs()
        save_df_to_npz(replicate_params, self.paths["nmf_replicate_parameters"])
        with open(self.paths["nmf_run_parameters"], "w") as F:
            yaml.dump(run_params, F)

    def prepare_from_matrix(self, norm_counts, components, n_iter=100, seed=None, beta_loss="frobenius",
                            alpha_usage=0.0, alpha_spectra=0.0, init="random", max_NMF_iter=1000,
                            tpm=None, _zero_cells_checked=False):
        """Stand-in for the tail of ``prepare`` (cnmf.py:452-459): persist the already
        normalised cells x HVG matrix (DataFrame or ndarray) and write the restart ledger +
        run parameters exactly as the reference does.  Raises the reference's zero-count
        error (cnmf.py:551-554).

        The object keeps REFERENCES to ``norm_counts`` and to the arrays of a sparse ``tpm`` (so that factorize / consensus of
        this process need not read back what was just written: 0.8 GB at 50 000 x 2 000): do not modify them in place
        afterwards -- the files on disk would no longer be what this process computes on.  Pass copies if you must."""
        import scipy.sparse as _sp
        sparse_in = None
        if isinstance(norm_counts, tuple) and len(norm_counts) == 3 and _sp.issparse(norm_counts[0]):
            sparse_in = norm_counts                              # (matrix, cell names, gene names)
        elif _sp.issparse(norm_counts):
            sparse_in = (norm_counts, ["cell%d" % i for i in range(norm_counts.shape[0])],
                         ["gene%d" % j for j in range(norm_counts.shape[1])])
        if sparse_in is not None:
            # the reference's sparse branch (cnmf.py:537-539, 550-556): norm_counts.X stays a scipy.sparse matrix and is
            # handed to scikit-learn as stored; here it is uploaded as CSR and walked on its stored entries by the
            # Kullback-Leibler paths (a coordinate-descent run forms the dense image on the device)
            mat = _sp.csr_matrix(sparse_in[0])
            if mat.dtype not in (np.float32, np.float64):
                mat = mat.astype(np.float64)
            if not mat.has_canonical_format:
                mat = mat.copy()
                mat.sum_duplicates()
            norm_counts = SparseFrame(mat, [str(c) for c in sparse_in[1]], [str(g) for g in sparse_in[2]])
            zerocells = np.asarray(mat.sum(axis=1)).ravel() == 0
        else:
            if not isinstance(norm_counts, pd.DataFrame):
                norm_counts = pd.DataFrame(np.asarray(norm_counts),
                                           index=["cell%d" % i for i in range(np.shape(norm_counts)[0])],
                                           columns=["gene%d" % j for j in range(np.shape(norm_counts)[1])])
            # (prepare_from_counts has the row sums from the device already: no second pass over 8 N G bytes on the host)
            zerocells = (np.zeros(norm_counts.shape[0], dtype=bool) if _zero_cells_checked
                         else np.array(norm_counts.values.sum(axis=1) == 0).reshape(-1))
        if zerocells.sum() > 0:
            examples = norm_counts.index[np.ravel(zerocells)]
            raise Exception("Error: %d cells have zero counts of overdispersed genes. E.g. %s. Filter those cells "
                            "and re-run or adjust the number of overdispersed genes. Quitting!"
                            % (zerocells.sum(), ", ".join(map(str, examples[:4]))))
        self._initialize_dirs()
        self._forget_results()
        sp_path = self.paths["normalized_counts_sparse"]
        for stale in (self.paths["normalized_counts"], self.paths["normalized_counts"] + ".data.npy", sp_path,
                      sp_path + ".cells.txt", sp_path + ".genes.txt") + tuple("%s.%s.npy" % (sp_path, part) for part in ("data", "indices", "indptr")):
            if os.path.exists(stale):
                os.remove(stale)                                 # (one form of the matrix on disk, never a stale other one)
        if sparse_in is not None:
            save_csr_fast(sp_path, norm_counts.values)
            with open(sp_path + ".cells.txt", "w") as F:
                F.write("\n".join(norm_counts.index))
            with open(sp_path + ".genes.txt", "w") as F:
                F.write("\n".join(norm_counts.columns))
        else:
            save_df_to_npz_fast(norm_counts, self.paths["normalized_counts"], sibling_ok=True)
        # this process holds what it just wrote: factorize() / consensus() need not read the 8 N G bytes back
        self._norm_counts_cache = ((self._nc_path(), os.path.getmtime(self._nc_path())), norm_counts)
        with open(self.paths["nmf_genes_list"], "w") as F:
            F.write("\n".join(map(str, norm_counts.columns)))
        for stale in (self.paths["tpm"], self.paths["tpm_sparse"], self.paths["tpm_sparse_genes"]) + tuple(
                "%s.%s.npy" % (self.paths["tpm_sparse"], part) for part in ("data", "indices", "indptr")):
            if tpm is not None and os.path.exists(stale):
                os.remove(stale)
        if tpm is not None and isinstance(tpm, tuple):
            # (scipy sparse cells x ALL genes, gene names): the reference keeps a sparse tpm.h5ad (cnmf.py:423-447);
            # statistics as get_mean_var does for sparse input (cnmf.py:126-134: population variance)
            import scipy.sparse as sp
            mat, genes = tpm
            mat = sp.csr_matrix(mat)
            if not mat.has_canonical_format:                   # (never in place on the caller's arrays)
                mat = mat.copy()
                mat.sum_duplicates()
            save_csr_fast(self.paths["tpm_sparse"], mat)
            with open(self.paths["tpm_sparse_genes"], "w") as F:
                F.write("\n".join(map(str, genes)))
            # (this process holds what it just wrote: consensus() need not read and CRC-check it back)
            self._tpm_sparse_cache = ((self.paths["tpm_sparse"], os.path.getmtime(self.paths["tpm_sparse"])),
                                      mat, pd.Index([str(g) for g in genes]))
            # column mean and E[x^2] in float64 as two weighted bin counts over the stored entries (row-major order, like
            # the sparse sums of get_mean_var; 3-10 x faster than .mean() / .multiply().mean() on a 50 000 x 2 000 matrix)
            d64 = mat.data.astype(np.float64)
            n_rows, n_cols = mat.shape
            mean = np.bincount(mat.indices, weights=d64, minlength=n_cols) / n_rows
            var = np.bincount(mat.indices, weights=d64 * d64, minlength=n_cols) / n_rows - mean ** 2
            stats = pd.DataFrame([mean, np.sqrt(np.maximum(var, 0.0))], index=["__mean", "__std"], columns=list(genes)).T
            save_df_to_npz(stats, self.paths["tpm_stats"])
        elif tpm is not None:
            save_df_to_npz(tpm, self.paths["tpm"])
            stats = pd.DataFrame([tpm.values.mean(axis=0), tpm.values.std(axis=0, ddof=0)],
                                 index=["__mean", "__std"], columns=tpm.columns).T
            save_df_to_npz(stats, self.paths["tpm_stats"])
        replicate_params, run_params = self.get_nmf_iter_params(
            ks=components, n_iter=n_iter, random_state_seed=seed, beta_loss=beta_loss,
            alpha_usage=alpha_usage, alpha_spectra=alpha_spectra, init=init, max_iter=max_NMF_iter)
        self.save_nmf_iter_params(replicate_params, run_params)

    def prepare_from_counts(self, counts, components, n_iter=100, seed=None, beta_loss="frobenius",
                            alpha_usage=0.0, alpha_spectra=0.0, init="random", max_NMF_iter=1000, tpm=None):
        """``get_norm_counts`` + the tail of ``prepare`` (cnmf.py:540-556, 452-459) with the
        normalisation on the device: ``counts`` holds the RAW counts of the chosen high-variance genes
        (cells x HVGs, DataFrame or ndarray; HVG selection itself stays with the reference / the
        caller).  The matrix is uploaded once, scaled to unit variance per gene (ddof=1, float64
        statistics), checked for zero cells, and STAYS resident for ``factorize``; the normalised
        matrix is written to the reference's ``normalized_counts`` file for the other workers."""
        if not isinstance(counts, pd.DataFrame):
            counts = pd.DataFrame(np.asarray(counts),
                                  index=["cell%d" % i for i in range(np.shape(counts)[0])],
                                  columns=["gene%d" % j for j in range(np.shape(counts)[1])])
        eng = self.engine
        eng.set_matrix(np.ascontiguousarray(counts.values, dtype=np.float32))
        self._engine_key = None
        self._resident_obj = None
        _, row_sums = eng.scale_genes_unit_variance()
        zerocells = row_sums == 0
        if zerocells.sum() > 0:
            examples = counts.index[np.ravel(zerocells)]
            raise Exception("Error: %d cells have zero counts of overdispersed genes. E.g. %s. Filter those cells "
                            "and re-run or adjust the number of overdispersed genes. Quitting!"
                            % (zerocells.sum(), ", ".join(map(str, examples[:4]))))
        norm_counts = pd.DataFrame(eng.get_matrix().astype(np.float64), index=counts.index, columns=counts.columns)
        # the init scale must be the one every OTHER worker derives from the file (set_matrix: X.mean() of the
        # float64 matrix), bit for bit -- not the device's own row-sum total
        x_mean, x_dtype = norm_counts.values.mean(), eng.x_dtype
        self.prepare_from_matrix(norm_counts, components, n_iter=n_iter, seed=seed, beta_loss=beta_loss,
                                 alpha_usage=alpha_usage, alpha_spectra=alpha_spectra, init=init,
                                 max_NMF_iter=max_NMF_iter, tpm=tpm, _zero_cells_checked=True)
        # the matrix just written is the one already resident: factorize() in this process skips the upload
        self._engine_key = ("norm_counts", self._nc_path(), os.path.getmtime(self._nc_path()))
        eng.x_mean, eng.x_dtype = x_mean, x_dtype
        return norm_counts

    # ------------------------------------------------------------------ the NMF call-site
    def _check_kwargs(self, kw):
        """The two solver configurations the reference can produce (cnmf.py:618-631):
        ('cd','frobenius') and ('mu', 'kullback-leibler' | 'itakura-saito')."""
        solver, beta = kw.get("solver", "cd"), kw.get("beta_loss", "frobenius")
        ok = (solver == "cd" and beta in ("frobenius", 2)) or \
             (solver == "mu" and beta in ("kullback-leibler", "itakura-saito", 1, 0))
        if not ok:
            raise NotImplementedError("the device engine implements solver='cd'/beta_loss='frobenius' and "
                                      "solver='mu'/beta_loss in ('kullback-leibler','itakura-saito'); "
                                      "got solver=%r beta_loss=%r" % (solver, beta))
        if kw.get("init", "random") not in ("random", "custom", "nndsvd", None) and "H" not in kw:
            raise NotImplementedError("init=%r is not implemented on the device (random / nndsvd / custom)" % kw.get("init"))

    def _nmf(self, X, nmf_kwargs):
        """Mirror of cNMF._nmf (cnmf.py:661-674): ``(spectra, usages)`` for one restart, or the
        NNLS refit when ``update_H=False`` (then ``H`` must have X's dtype, sklearn _nmf.py:1221)."""
        kw = dict(nmf_kwargs)
        self._check_kwargs(kw)
        Xv = X.values if isinstance(X, (pd.DataFrame, SparseFrame)) else X
        # (the matrix object consensus()/factorize() made resident is not uploaded again; anything else is)
        if self._resident_obj is not None and X is self._resident_obj and self._engine is not None:
            eng = self._engine
        else:
            eng = self._get_engine(Xv, None)
        mu = kw.get("solver", "cd") == "mu"
        if kw.get("update_H", True) is False:
            H = np.asarray(kw["H"])
            xdt = Xv.dtype if Xv.dtype in (np.float32, np.float64) else np.dtype(np.float64)
            if H.dtype != xdt:
                raise TypeError("H should have the same dtype as X. Got H.dtype = {}.".format(H.dtype))
            if mu:
                # float64 on the stored entries, like scikit-learn on the reference's float64 matrices (cnmf.py:534); both
                # beta losses (Itakura-Saito: a strictly positive matrix -- scikit-learn's rule -- stores every entry)
                W, _, _ = eng.mu_refit_f64(H, tol=kw.get("tol", 1e-4), max_iter=kw.get("max_iter", 200),
                                           alpha_W=kw.get("alpha_W", 0.0), l1_ratio=kw.get("l1_ratio", 0.0),
                                           beta_loss=kw["beta_loss"])
            else:
                # scikit-learn solves in X's dtype (sklearn _nmf.py:1221-1233): float64 matrices get the float64 refit
                # (product, Gram matrix and sweeps), float32 ones the float32 matrix-pipe path
                refit = eng.nnls_f64 if xdt == np.float64 else eng.nnls
                W, _ = refit(H, tol=kw.get("tol", 1e-4), max_iter=kw.get("max_iter", 200),
                             alpha_W=kw.get("alpha_W", 0.0), l1_ratio=kw.get("l1_ratio", 0.0))
            return H, W.astype(xdt, copy=False)
        k = int(kw["n_components"])
        if kw.get("init") == "nndsvd":                       # `--init nndsvd` (cnmf.py:1252)
            W0, H0 = eng.nndsvd_init(k, random_state=int(kw["random_state"]))
            kw = dict(kw, init="custom", W=W0, H=H0)
        if mu:
            common = dict(beta_loss=kw["beta_loss"], tol=kw.get("tol", 1e-4), max_iter=kw.get("max_iter", 200),
                          alpha_W=kw.get("alpha_W", 0.0), alpha_H=kw.get("alpha_H", 0.0),
                          l1_ratio=kw.get("l1_ratio", 0.0), return_W=True)
            if kw.get("init") == "custom":
                Hl, Wl, _, _ = eng.nmf_mu_batch([k], W0=[kw["W"]], H0=[kw["H"]], **common)
            else:
                Hl, Wl, _, _ = eng.nmf_mu_batch([k], seeds=[int(kw["random_state"])], **common)
            xdt = Xv.dtype if Xv.dtype in (np.float32, np.float64) else np.dtype(np.float64)
            return Hl[0].astype(xdt), Wl[0].astype(xdt)
        if kw.get("init") == "custom":
            Hl, Wl, _, _ = eng.nmf_batch([k], W0=[kw["W"]], H0=[kw["H"]], tol=kw.get("tol", 1e-4),
                                         max_iter=kw.get("max_iter", 200), alpha_W=kw.get("alpha_W", 0.0),
                                         alpha_H=kw.get("alpha_H", 0.0), l1_ratio=kw.get("l1_ratio", 0.0),
                                         return_W=True)
        else:
            Hl, Wl, _, _ = eng.nmf_batch([k], seeds=[int(kw["random_state"])], tol=kw.get("tol", 1e-4),
                                         max_iter=kw.get("max_iter", 200), alpha_W=kw.get("alpha_W", 0.0),
                                         alpha_H=kw.get("alpha_H", 0.0), l1_ratio=kw.get("l1_ratio", 0.0),
                                         return_W=True)
        xdt = Xv.dtype if Xv.dtype in (np.float32, np.float64) else np.dtype(np.float64)
        return Hl[0].astype(xdt), Wl[0].astype(xdt)

    # ------------------------------------------------------------------ factorize (cnmf.py:692-745)
    def factorize(self, worker_i=0, total_workers=1, skip_completed_runs=False, write_iter_files=True,
                  kc_max=0, iteration_hints=None):
        """cnmf.py:692-745: ONE batched device call for this worker's ledger rows.  ``iteration_hints`` ({rank: expected
        outer iterations}, e.g. ``self.learned_iterations`` of an earlier factorize on the same data): the queue starts
        longest-expected-first (Engine.set_iteration_hints); results then differ from an unhinted call in the last bits."""
        import time as _time
        _t = [_time.perf_counter()]
        run_params = load_df_from_npz(self.paths["nmf_replicate_parameters"])
        norm_counts = self._load_norm_counts()
        _t.append(_time.perf_counter())
        _nmf_kwargs = yaml.load(open(self.paths["nmf_run_parameters"]), Loader=yaml.FullLoader)
        self._check_kwargs(_nmf_kwargs)
        if not skip_completed_runs:
            jobs = list(worker_filter(range(len(run_params)), worker_i, total_workers))
        else:
            jobs = list(worker_filter(run_params.index[run_params["completed"] == False],   # noqa: E712
                                      worker_i, total_workers))
        self.last_factorize_jobs = [int(j) for j in jobs]     # ledger rows THIS call ran (dist.factorize_distributed)
        if not jobs:
            return
        eng = self._get_engine(norm_counts.values, ("norm_counts", self._nc_path(), os.path.getmtime(self._nc_path())))
        _t.append(_time.perf_counter())
        sub = run_params.iloc[jobs]
        ks = [int(v) for v in sub["n_components"].values]
        seeds = [int(v) for v in sub["nmf_seed"].values]
        its = [int(v) for v in sub["iter"].values]
        print("\n".join("[Worker %d]. Starting task %d." % (worker_i, idx) for idx in jobs))
        common = dict(tol=_nmf_kwargs.get("tol", 1e-4), max_iter=_nmf_kwargs.get("max_iter", 1000),
                      alpha_W=_nmf_kwargs.get("alpha_W", 0.0), alpha_H=_nmf_kwargs.get("alpha_H", 0.0),
                      l1_ratio=_nmf_kwargs.get("l1_ratio", 0.0))
        init_kw = dict(seeds=seeds)
        if _nmf_kwargs.get("init") == "nndsvd":
            inits = eng.nndsvd_init_batch(ks, seeds)           # range finders of up to 13 restarts per pass over X
            init_kw = dict(W0=[w for w, _ in inits], H0=[h for _, h in inits])
        if _nmf_kwargs.get("solver", "cd") == "mu":
            H_list, _, n_iter, _ = eng.nmf_mu_batch(ks, beta_loss=_nmf_kwargs["beta_loss"], **init_kw, **common)
            self.last_factorize_stats = dict(n_iter=n_iter)
        else:
            # the spectra also stay in the engine's device store: k selection and consensus of THIS process take their
            # merged spectra from there (no 80 MB upload per consensus call; round-3 review, next #9)
            keep = hasattr(eng, "spectra_fetch")
            if keep and self._store_rows and all(key in set(zip(ks, its)) for key in self._store_rows):
                # every restart this object keeps in the device store is about to be run again: drop the old rows instead
                # of appending behind them (a bench loop / a repeated factorize would grow the store by 65 MB per call)
                eng.spectra_reset()
                self._store_rows.clear()
            if iteration_hints is not None:
                eng.set_iteration_hints(iteration_hints)
            try:
                H_list, _, n_iter, _ = eng.nmf_batch(ks, kc_max=kc_max, resident="keep" if keep else False, **init_kw, **common)
            finally:
                if iteration_hints is not None:
                    eng.set_iteration_hints(None)
            self.last_factorize_stats = dict(eng.last_stats, n_iter=n_iter)
            if hasattr(eng, "iteration_means"):
                self.learned_iterations = eng.iteration_means()
            if keep:
                for k, it, off in zip(ks, its, eng.last_store_offsets):
                    self._store_rows[(k, it)] = (int(off), eng.last_store_gen)
        _t.append(_time.perf_counter())
        xdt = norm_counts.values.dtype if norm_counts.values.dtype in (np.float32, np.float64) else np.float64
        self._spectra_columns = norm_counts.columns
        for k, it, H in zip(ks, its, H_list):
            # kept as plain arrays (k x genes, X's dtype): 900 DataFrames cost more host time than they are worth;
            # combine_nmf() builds ONE frame per k
            arr = H.astype(xdt)
            self.spectra_cache[(k, it)] = arr
            if write_iter_files:
                save_df_to_npz(pd.DataFrame(arr, index=np.arange(1, k + 1), columns=norm_counts.columns),
                               self.paths["iter_spectra"] % (k, it))
        _t.append(_time.perf_counter())
        self.last_factorize_stats["host_seconds"] = dict(load_inputs=_t[1] - _t[0], upload=_t[2] - _t[1],
                                                         device_call=_t[3] - _t[2], store_results=_t[4] - _t[3])

    def factorize_multi_gpu(self, n_gpus=None, skip_completed_runs=False, gather="rccl", **kw):
        """``factorize`` on ``n_gpus`` GPUs of this node, one process per GPU, and the gather: afterwards the merged
        spectra files of every k exist, as after ``factorize`` + ``combine`` (cnmf_amd/dist.py::factorize_multi_gpu)."""
        from . import dist
        return dist.factorize_multi_gpu(self, n_gpus=n_gpus, skip_completed_runs=skip_completed_runs, gather=gather, **kw)

    def factorize_multi_process(self, total_workers, skip_completed_runs=False):
        """The reference's entry point (cnmf.py:677-689): ``total_workers`` worker processes, each running
        ``factorize(worker_i, total_workers)`` and writing its per-iteration files -- here one worker per GPU
        (``total_workers`` must not exceed the GPUs of the node), gathered through the files like the reference;
        ``combine()`` stays the caller's next step, as in the reference."""
        from . import _lib
        n_dev = int(_lib.load().cnmf_device_count())
        if int(total_workers) > max(n_dev, 1):
            raise ValueError("total_workers=%d exceeds the %d GPU(s) of this node (one worker process per GPU)"
                             % (int(total_workers), n_dev))
        from . import dist
        argv_obj = self
        return dist.factorize_multi_gpu(argv_obj, n_gpus=int(total_workers), skip_completed_runs=skip_completed_runs,
                                        gather="files")

    # ------------------------------------------------------------------ combine (cnmf.py:462-483, 748-773)
    def combine(self, components=None, skip_missing_files=False):
        if type(components) is int:
            ks = [components]
        elif components is None:
            run_params = load_df_from_npz(self.paths["nmf_replicate_parameters"])
            ks = sorted(set(run_params.n_components))
        else:
            ks = components
        for k in ks:
            self.combine_nmf(k, skip_missing_files=skip_missing_files)

    def combine_nmf(self, k, skip_missing_files=False, remove_individual_iterations=False):
        run_params = load_df_from_npz(self.paths["nmf_replicate_parameters"])
        print("Combining factorizations for k=%d." % k)
        run_params_subset = run_params[run_params.n_components == k].sort_values("iter")
        blocks, labels, columns = [], [], None
        for kk, itv in zip(run_params_subset["n_components"].values, run_params_subset["iter"].values):
            key = (int(kk), int(itv))
            current_file = self.paths["iter_spectra"] % key
            if key in self.spectra_cache:
                block = self.spectra_cache[key]
                if columns is None:
                    columns = self._spectra_columns
                elif block.shape[1] != len(columns):
                    raise ValueError("restart %r in memory has %d genes, the other restarts of k=%d have %d"
                                     % (key, block.shape[1], k, len(columns)))
            elif os.path.exists(current_file):
                df = load_df_from_npz(current_file)
                if columns is None:
                    columns = df.columns
                elif not (len(df.columns) == len(columns) and (np.asarray(df.columns) == np.asarray(columns)).all()):
                    # (a stale file of another gene selection in the same output directory; the reference's pd.concat
                    #  would have aligned by name and padded with NaN -- here the blocks are stacked as arrays)
                    if set(df.columns) != set(columns):
                        raise ValueError("%s holds spectra over a different gene set than the other restarts of k=%d "
                                         "(stale file of an earlier prepare?)" % (current_file, k))
                    df = df.loc[:, list(columns)]
                block = df.values
            else:
                if not skip_missing_files:
                    print("Missing file: %s, run with skip_missing=True to override" % current_file)
                    raise FileNotFoundError(errno.ENOENT, os.strerror(errno.ENOENT), current_file)
                print("Missing file: %s. Skipping." % current_file)
                continue
            blocks.append(block)
            labels.extend("iter%d_topic%d" % (itv, t + 1) for t in range(k))
        combined_spectra = []
        if len(blocks) > 0:
            combined_spectra = pd.DataFrame(np.concatenate(blocks, axis=0), index=labels, columns=columns)
            (save_df_to_npz if self.compress_merged else save_df_to_npz_fast)(combined_spectra, self.paths["merged_spectra"] % k)
            self.merged_cache[k] = (os.path.getmtime(self.paths["merged_spectra"] % k), combined_spectra)
            if remove_individual_iterations:
                for i, p in run_params_subset.iterrows():
                    f = self.paths["iter_spectra"] % (int(p["n_components"]), int(p["iter"]))
                    if os.path.exists(f):
                        os.remove(f)
        else:
            print("No spectra found for k=%d" % k)
        return combined_spectra

    # ------------------------------------------------------------------ refits (cnmf.py:776-820)
    def refit_usage(self, X, spectra):
        refit_nmf_kwargs = yaml.load(open(self.paths["nmf_run_parameters"]), Loader=yaml.FullLoader)
        Hv = spectra.values if type(spectra) is pd.DataFrame else spectra
        refit_nmf_kwargs.update(dict(n_components=Hv.shape[0], H=Hv, update_H=False))
        _, rf_usages = self._nmf(X, nmf_kwargs=refit_nmf_kwargs)
        if (type(X) is pd.DataFrame) and (type(spectra) is pd.DataFrame):
            rf_usages = pd.DataFrame(rf_usages, index=X.index, columns=spectra.index)
        return rf_usages

    def refit_spectra(self, X, usage):
        return self.refit_usage(X.T, usage.T).T

    def _merged_store_rows(self, k, merged_index, eng):
        """Row indices into ``eng``'s resident spectra store for the merged spectra of ``k`` (labels iter%d_topic%d), or
        None when any of them is not there (another process ran it, the store was reset, the merged file was not written
        by this process's combine)."""
        cached = self.merged_cache.get(k)
        if (cached is None or eng is not self._engine or not self._store_rows or not hasattr(eng, "store_gen")
                or cached[0] != os.path.getmtime(self.paths["merged_spectra"] % k)):
            return None
        rows = np.empty(len(merged_index), dtype=np.int64)
        pos = 0
        while pos < len(merged_index):
            label = merged_index[pos]
            try:
                it = int(label[4:label.index("_topic")])
            except (ValueError, TypeError):
                return None
            ent = self._store_rows.get((k, it))
            if ent is None or ent[1] != eng.store_gen or pos + k > len(merged_index) or merged_index[pos + k - 1] != "iter%d_topic%d" % (it, k):
                return None
            rows[pos:pos + k] = ent[0] + np.arange(k)
            pos += k
        return rows

    # ------------------------------------------------------------------ consensus (cnmf.py:823-985)
    def consensus(self, k, density_threshold=0.5, local_neighborhood_size=0.30, show_clustering=True,
                  build_ref=True, skip_density_and_return_after_stats=False, close_clustergram_fig=False,
                  refit_usage=True, normalize_tpm_spectra=False, norm_counts=None):
        """cnmf.py:823-985 without the plotting block and ``build_reference`` (out of scope, SURVEY section 2
        #10/#11): same signature and defaults as the reference, so positional (cnmf.py:1290) and keyword-less
        calls bind alike.  With ``show_clustering=True`` AND ``self.materialize_topics_dist`` the distance matrix
        is fetched as ``self.topics_dist`` for the reference's unchanged plotting code (integration/hip_backend.py);
        ``build_ref`` and ``close_clustergram_fig`` have nothing to act on here and are ignored."""
        cached = self.merged_cache.get(k)
        if cached is not None and cached[0] == os.path.getmtime(self.paths["merged_spectra"] % k):
            merged_spectra = cached[1].copy()                # this process wrote that very file
        else:
            merged_spectra = load_df_from_npz(self.paths["merged_spectra"] % k)
        if norm_counts is None:
            norm_counts = self._load_norm_counts()
        density_threshold_str = str(density_threshold)
        if skip_density_and_return_after_stats:
            density_threshold_str = "2"
        density_threshold_repl = density_threshold_str.replace(".", "_")
        nc_key = ("norm_counts", self._nc_path(), os.path.getmtime(self._nc_path()))
        eng = self._get_engine(norm_counts.values, nc_key)
        self._resident_obj = norm_counts                      # the refit below reuses this upload
        R = merged_spectra.shape[0]
        n_neighbors = int(local_neighborhood_size * R / k)    # cnmf.py:879
        # local-density cache (cnmf.py:887-899).  The reference reuses the file whenever it exists -- also after
        # `local_neighborhood_size` or the merged spectra changed.  Here the file is reused when it is the
        # reference's own (no side-car) or when the side-car written next to it names the same neighbourhood
        # and spectra count; otherwise the density is recomputed and the file refreshed.
        cache_path = self.paths["local_density_cache"] % k
        meta_path = cache_path + ".meta.json"
        cached_density = None
        if not skip_density_and_return_after_stats and os.path.isfile(cache_path):
            ok = True
            if os.path.isfile(meta_path):
                import json
                try:
                    meta = json.load(open(meta_path))
                    ok = meta.get("n_neighbors") == n_neighbors and meta.get("n_spectra") == R
                except ValueError:
                    ok = False
            if ok:
                ld = load_df_from_npz(cache_path)
                if ld.shape[0] == R:
                    cached_density = np.asarray(ld.iloc[:, 0].values, dtype=np.float64)
        if cached_density is not None:
            keep = cached_density < density_threshold          # strict <, cnmf.py:903
            if keep.sum() == 0:
                raise RuntimeError("Zero components remain after density filtering. Consider increasing density threshold")
            srows = self._merged_store_rows(k, merged_spectra.index, eng)
            sub = eng.consensus(None if srows is not None else merged_spectra.values[keep], k, skip_density=True,
                                want_silhouette=False, return_dist=False,
                                **({"store_rows": srows[keep]} if srows is not None else {}))
            labels = -np.ones(R, dtype=np.int32)
            labels[keep] = sub["labels"]
            out = dict(sub, local_density=cached_density, density_filter=keep, labels=labels)
            if show_clustering:
                out["topics_dist"] = None      # like the reference with a cached density (cnmf.py:886, 988-990)
        else:
            srows = self._merged_store_rows(k, merged_spectra.index, eng)
            out = eng.consensus(None if srows is not None else merged_spectra.values, k, density_threshold=density_threshold,
                                **({"store_rows": srows} if srows is not None else {}),
                                local_neighborhood_size=local_neighborhood_size,
                                skip_density=skip_density_and_return_after_stats,
                                want_silhouette=skip_density_and_return_after_stats,
                                return_dist=(show_clustering and self.materialize_topics_dist
                                             and not skip_density_and_return_after_stats))
            if not skip_density_and_return_after_stats:
                import json
                local_density = pd.DataFrame(out["local_density"], columns=["local_density"], index=merged_spectra.index)
                save_df_to_npz(local_density, cache_path)
                with open(meta_path, "w") as F:
                    json.dump({"n_neighbors": n_neighbors, "n_spectra": int(R),
                               "local_neighborhood_size": float(local_neighborhood_size)}, F)
        # artefacts the (unchanged) plotting block of the reference consumes (cnmf.py:986-1079)
        self.topics_dist = out.get("topics_dist")
        self.density_filter = pd.Series(out["density_filter"], index=merged_spectra.index)
        kept = out["density_filter"]
        self.kmeans_cluster_labels = pd.Series(out["labels"][kept] + 1, index=merged_spectra.index[kept])
        self.local_density = out["local_density"]
        median_spectra = pd.DataFrame(out["median_spectra"], index=np.arange(1, k + 1), columns=merged_spectra.columns)
        xdt = norm_counts.values.dtype
        rf_usages = self.refit_usage(norm_counts, median_spectra.astype(xdt))
        rf_usages = pd.DataFrame(np.asarray(rf_usages), index=norm_counts.index, columns=median_spectra.index)

        if skip_density_and_return_after_stats:
            prediction_error = eng.prediction_error(rf_usages.values, median_spectra.values)
            return pd.DataFrame([k, density_threshold, out["silhouette"], prediction_error],
                                index=["k", "local_density_threshold", "silhouette", "prediction_error"],
                                columns=["stats"])

        # re-order by total contribution (cnmf.py:939-946)
        norm_usages = rf_usages.div(rf_usages.sum(axis=1), axis=0)
        reorder = norm_usages.sum(axis=0).sort_values(ascending=False)
        rf_usages = rf_usages.loc[:, reorder.index]
        norm_usages = norm_usages.loc[:, reorder.index]
        median_spectra = median_spectra.loc[reorder.index, :]
        rf_usages.columns = np.arange(1, rf_usages.shape[1] + 1)
        norm_usages.columns = rf_usages.columns
        median_spectra.index = rf_usages.columns

        spectra_tpm = usage_coef = None
        have_dense, have_sparse = os.path.exists(self.paths["tpm"]), os.path.exists(self.paths["tpm_sparse"])
        if have_dense or have_sparse:
            # consensus tail (cnmf.py:948-975) on the device.  The TPM matrix (cells x ALL genes, dense or CSR) is
            # uploaded ONCE and serves all three steps:
            #   refit_spectra  -> NNLS over the gene rows with the product W^T.X (no transposed upload)
            #   OLS z-scores   -> X^T Y with Y z-scored on the fly, float64 accumulation (cnmf.py:55-125)
            #   final refit    -> usages on tpm[:, hvgs] / std as a product with the resident matrix
            if have_sparse:
                import scipy.sparse as sp
                key = (self.paths["tpm_sparse"], os.path.getmtime(self.paths["tpm_sparse"]))
                cached = getattr(self, "_tpm_sparse_cache", None)
                if cached is not None and cached[0] == key:
                    tpm_x, tpm_genes = cached[1], cached[2]
                else:
                    tpm_x = load_csr(self.paths["tpm_sparse"])
                    tpm_genes = pd.Index(open(self.paths["tpm_sparse_genes"]).read().split("\n"))
            else:
                tpm = load_df_from_npz(self.paths["tpm"])
                tpm_x, tpm_genes = tpm.values, tpm.columns
            tpm_stats = load_df_from_npz(self.paths["tpm_stats"])
            tdt = tpm_x.dtype if tpm_x.dtype in (np.float32, np.float64) else np.dtype(np.float64)
            eng = self._get_engine(tpm_x, None)
            kw = yaml.load(open(self.paths["nmf_run_parameters"]), Loader=yaml.FullLoader)
            solver_kw = dict(tol=kw.get("tol", 1e-4), max_iter=kw.get("max_iter", 1000),
                             alpha_W=kw.get("alpha_W", 0.0), l1_ratio=kw.get("l1_ratio", 0.0))
            mu_tail = kw.get("solver", "cd") == "mu"
            if mu_tail:
                # refit_spectra = the multiplicative-update refit of tpm.X^T against usages^T (cnmf.py:805-820): float64 on the
                # compressed rows of X^T, built on the device from the resident matrix -- no todense(), no transposed upload
                # (Kullback-Leibler: the stored entries; Itakura-Saito, round 6: the matrix is strictly positive by
                # scikit-learn's rule, every entry is stored)
                solver_kw["beta_loss"] = kw.get("beta_loss")
                Wt, _, _ = eng.mu_refit_f64(norm_usages.values.T, transposed=True, **solver_kw)
                spectra_tpm = Wt.T
            else:
                spectra_tpm, _ = eng.nnls_spectra(norm_usages.values.astype(tdt), **solver_kw)
            spectra_tpm = pd.DataFrame(np.asarray(spectra_tpm, dtype=tdt), index=rf_usages.columns, columns=tpm_genes)
            if normalize_tpm_spectra:
                spectra_tpm = spectra_tpm.div(spectra_tpm.sum(axis=1), axis=0) * 1e6
            # z-score spectra: Beta = lstsq(X^T X, X^T Y) with Y = (tpm - mean) / std, var floored at 1e-12
            mean, pvar = eng.col_mean_var()                 # float64 column mean / population variance
            var = np.where(pvar < 1e-12, 1e-12, pvar)
            Xd = rf_usages.values.astype(np.float64)
            XtY = eng.xt_matmul_f64(Xd, mean=mean, std=np.sqrt(var))
            usage_coef, *_ = np.linalg.lstsq(Xd.T @ Xd, XtY, rcond=None)
            usage_coef = pd.DataFrame(usage_coef, index=rf_usages.columns, columns=tpm_genes)
            if refit_usage:
                hvgs = open(self.paths["nmf_genes_list"]).read().split("\n")
                hidx = tpm_genes.get_indexer(hvgs)
                if (hidx < 0).any():
                    raise KeyError("high-variance genes missing from the TPM matrix")
                n_cells = eng.shape[0]
                # std with ddof=1 of the HVG columns (dense: X.std(ddof=1); sparse: sc.pp.scale(zero_center=False))
                std1 = np.sqrt(pvar[hidx] * n_cells / (n_cells - 1.0))
                spectra_tpm_rf = spectra_tpm.loc[:, hvgs].div(tpm_stats.loc[hvgs, "__std"], axis=1)
                Hrf = spectra_tpm_rf.values.astype(np.float64)
                H_prod = np.zeros((Hrf.shape[0], len(tpm_genes)), dtype=np.float64)
                H_prod[:, hidx] = Hrf / std1
                if mu_tail:
                    # the same refit on tpm[:, hvgs] / std without forming that matrix: the resident TPM's entries divided
                    # by the gene's std on the fly, the other columns dropped; W0 = sqrt(mean(tpm[:, hvgs] / std) / k)
                    div = np.zeros(len(tpm_genes))
                    div[hidx] = std1
                    H_full = np.zeros((Hrf.shape[0], len(tpm_genes)))
                    H_full[:, hidx] = Hrf
                    w0 = float(np.sqrt((mean[hidx] / std1).mean() / Hrf.shape[0]))
                    rf, _, _ = eng.mu_refit_f64(H_full, col_divisor=div, w_init=w0, n_features=len(hvgs), **solver_kw)
                elif tdt == np.float64:
                    rf, _ = eng.nnls_f64(H_prod, gram=Hrf @ Hrf.T, n_features=len(hvgs), **solver_kw)
                else:
                    rf, _ = eng.nnls_gram(H_prod, Hrf @ Hrf.T, n_features=len(hvgs), **solver_kw)
                rf_usages = pd.DataFrame(np.asarray(rf, dtype=xdt), index=norm_counts.index, columns=spectra_tpm_rf.index)

        # the nine artefacts of cnmf.py:977-985, written side by side: zlib and the float formatting of the two large
        # tables (usages: cells x k) overlap instead of adding up
        rep = (k, density_threshold_repl)
        writes = [(save_df_to_npz, median_spectra, self.paths["consensus_spectra"] % rep),
                  (save_df_to_npz, rf_usages, self.paths["consensus_usages"] % rep),
                  (save_df_to_text, median_spectra, self.paths["consensus_spectra__txt"] % rep),
                  (save_df_to_text, rf_usages, self.paths["consensus_usages__txt"] % rep)]
        if spectra_tpm is not None:
            writes += [(save_df_to_npz, spectra_tpm, self.paths["gene_spectra_tpm"] % rep),
                       (save_df_to_text, spectra_tpm, self.paths["gene_spectra_tpm__txt"] % rep),
                       (save_df_to_npz, usage_coef, self.paths["gene_spectra_score"] % rep),
                       (save_df_to_text, usage_coef, self.paths["gene_spectra_score__txt"] % rep)]
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=4) as pool:
            for fut in [pool.submit(fn, obj, path) for fn, obj, path in writes]:
                fut.result()                                   # a failed write raises here, as it would have inline
        return median_spectra, rf_usages

    # ------------------------------------------------------------------ k selection (cnmf.py:1119-1135)
    def k_selection_stats(self, batched=True):
        """The numerical half of ``k_selection_plot`` (cnmf.py:1119-1135); writes ``k_selection_stats.df.npz`` like the
        reference (plotting is out of scope).  ``batched=True`` (default): ONE device call for all k -- the norm counts
        are uploaded once, the |K| stats-mode consensuses run back to back, their usage refits as one batched NNLS
        (all spectra are columns of a single X.H^T pass) and the prediction errors with the usages still on the device.
        ``batched=False``: the reference's loop, one ``consensus(k, skip_density_and_return_after_stats=True)`` per k."""
        run_params = load_df_from_npz(self.paths["nmf_replicate_parameters"])
        norm_counts = self._load_norm_counts()
        ks = sorted(set(int(k) for k in run_params.n_components))
        kw = yaml.load(open(self.paths["nmf_run_parameters"]), Loader=yaml.FullLoader)
        if batched and kw.get("solver", "cd") == "cd":
            nc_key = ("norm_counts", self._nc_path(), os.path.getmtime(self._nc_path()))
            eng = self._get_engine(norm_counts.values, nc_key)
            self._resident_obj = norm_counts
            merged, srows = {}, {}
            for k in ks:
                cached = self.merged_cache.get(k)
                if cached is not None and cached[0] == os.path.getmtime(self.paths["merged_spectra"] % k):
                    rows = self._merged_store_rows(k, cached[1].index, eng)
                    if rows is not None:
                        srows[k] = rows
                    merged[k] = cached[1].values
                else:
                    merged[k] = load_df_from_npz(self.paths["merged_spectra"] % k).values
            solver_kw = dict(nnls_tol=kw.get("tol", 1e-4), nnls_max_iter=kw.get("max_iter", 1000),
                             alpha_W=kw.get("alpha_W", 0.0), l1_ratio=kw.get("l1_ratio", 0.0))
            if len(srows) == len(ks):          # every k still sits in the device store of this process: no upload at all
                res = eng.kselect_stats(None, store_rows_by_k=srows, **solver_kw)
            else:
                res = eng.kselect_stats(merged, **solver_kw)
            stats = pd.DataFrame([[k, 0.5, res[k]["silhouette"], res[k]["prediction_error"]] for k in ks],
                                 columns=["k", "local_density_threshold", "silhouette", "prediction_error"])
            stats["k"] = stats["k"].astype(float)
            stats.columns.name = None
        else:
            stats = []
            for k in ks:
                stats.append(self.consensus(k, skip_density_and_return_after_stats=True,
                                            show_clustering=False, close_clustergram_fig=True,
                                            norm_counts=norm_counts).stats)
            stats = pd.DataFrame(stats)
            stats.reset_index(drop=True, inplace=True)
        save_df_to_npz(stats, self.paths["k_selection_stats"])
        return stats

