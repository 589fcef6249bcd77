"""Golden vectors for the FULL-SIZE configs, produced by scikit-learn itself (float64) in the build container.

The GPU box has no time budget for hundreds of float64 outer iterations at 50 000 x 2000, so the
oracle runs here and its output is committed (`tests/golden/ref_c3_long.npz`, `ref_c4_csr.npz`);
`tests/test_gpu_golden_big.py` regenerates the same seeded inputs on the GPU box and compares.

C3 (north-star shape, 50 000 x 2000): for k = 5, 11, 13 the first ledger seed (cnmf.py:593-610, seed 14,
K = 5..13, n_iter = 100) whose run needs >= 300 outer iterations: spectra after exactly 50 and 150 iterations
(`max_iter`, the same truncation the device is asked for) and at the stopping rule (tol 1e-4, max_iter 1000), the
final objective, and -- as calibration of what float32 can deliver on these ill-conditioned trajectories -- the
drift of scikit-learn's own float32 path from its float64 path at the same truncations.
C4 (200 000 x 2000 CSR, ~8 % dense, K = 20): two restarts x 10 outer iterations on the sparse input -- once with gamma-
distributed values (the general GEMM path) and once COUNT-valued (Poisson counts / std: the default f16 count path).

    python tools/make_golden_big.py        # ~15 min on 8 cores
    python tools/make_golden_big.py c3drift      # only the float32-at-the-stopping-rule calibration (round 4)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import scipy.sparse as sp

from cnmf_amd import synth
from oracle import sklearn_ref

OUT = os.path.join(ROOT, "tests", "golden")


def c4_matrix():
    """The C4 input of tests/test_gpu_configs.py (same seeded construction)."""
    rs = np.random.RandomState(3)
    N, G = 200_000, 2000
    X = sp.random(N, G, density=0.08, format="csr", dtype=np.float32, random_state=rs,
                  data_rvs=lambda n: rs.gamma(1.0, 1.0, size=n).astype(np.float32))
    return X[np.asarray(X.sum(axis=1)).ravel() > 0]


def make_c3():
    from oracle import nmf_cd
    X32 = synth.make_config("C3", dtype=np.float32)
    X = X32.astype(np.float64)
    led = sklearn_ref.ledger(list(range(5, 14)), 100, 14)
    out = {"x_checksum": np.array([X.sum(), (X * X).sum()])}
    for k in (5, 11, 13):
        for (kk, it, seed) in led:
            if kk != k:
                continue
            t0 = time.time()
            H_full, W_full, n_full = sklearn_ref.nmf(X, k, seed)
            print("C3 k=%d iter=%d seed=%d: n_iter=%d (%.0f s)" % (k, it, seed, n_full, time.time() - t0), flush=True)
            if n_full < 300:
                continue
            out["k%d_seed" % k] = np.array([seed, it, n_full], dtype=np.int64)
            out["k%d_Hfull" % k] = H_full.astype(np.float32)
            out["k%d_objfull" % k] = np.array([((X - W_full @ H_full) ** 2).sum()])
            for T in (50, 150):
                H64, _, n = sklearn_ref.nmf(X, k, seed, max_iter=T)
                assert n == T
                out["k%d_H%d" % (k, T)] = H64.astype(np.float32)
                # calibration: how far scikit-learn's OWN float32 path drifts from its float64 path on this
                # restart at the same truncation (k != K_true = 9 trajectories are ill-conditioned: rounding-level
                # perturbations are amplified) -- the device is held to a small multiple of this
                H32, _, _ = sklearn_ref.nmf(X32, k, seed, max_iter=T)
                dev = nmf_cd.spectra_error(H64, H32)
                out["k%d_f32dev%d" % (k, T)] = np.array(dev)
                print("   T=%d sklearn float32 vs float64: maxabs %.2e relfro %.2e" % (T, dev[0], dev[1]), flush=True)
            break
    np.savez_compressed(os.path.join(OUT, "ref_c3_long.npz"), **out)


def make_c3_drift():
    """Adds to ref_c3_long.npz how far scikit-learn's OWN float32 path lands from its float64 path AT THE STOPPING RULE
    on the three long restarts (round-3 review, weak #2: the device was held to a bare 5e-3 / 2e-2 there): spectra
    distance, iteration count and objective of the float32 run.  Leaves every existing entry untouched."""
    from oracle import nmf_cd
    path = os.path.join(OUT, "ref_c3_long.npz")
    out = dict(np.load(path))
    X32 = synth.make_config("C3", dtype=np.float32)
    X = X32.astype(np.float64)
    assert np.allclose([X.sum(), (X * X).sum()], out["x_checksum"], rtol=1e-12)
    for k in (5, 11, 13):
        seed, _, n_full = (int(v) for v in out["k%d_seed" % k])
        t0 = time.time()
        H32, W32, n32 = sklearn_ref.nmf(X32, k, seed)
        dev = nmf_cd.spectra_error(out["k%d_Hfull" % k].astype(np.float64), H32.astype(np.float64))
        obj32 = ((X - W32.astype(np.float64) @ H32.astype(np.float64)) ** 2).sum()
        out["k%d_f32devfull" % k] = np.array(dev)
        out["k%d_f32nfull" % k] = np.array([n32], dtype=np.int64)
        out["k%d_f32objfull" % k] = np.array([obj32])
        print("C3 k=%d seed=%d: float32 stops at %d (float64: %d), spectra maxabs %.2e relfro %.2e, objective %.8g vs %.8g (%.0f s)"
              % (k, seed, n32, n_full, dev[0], dev[1], obj32, float(out["k%d_objfull" % k][0]), time.time() - t0), flush=True)
    np.savez_compressed(path, **out)


def make_c3_nndsvd():
    """scikit-learn's init='nndsvd' (_initialize_nmf, sklearn/decomposition/_nmf.py:316-354) at the FULL C3 size
    (50 000 x 2000; round-3 review: the device NNDSVD was only tested up to 9 000 cells): H0 in full, W0 on every 97th
    cell plus its column sums and sums of squares over ALL cells."""
    from sklearn.decomposition._nmf import _initialize_nmf
    X = synth.make_config("C3", dtype=np.float32).astype(np.float64)
    out = {"x_checksum": np.array([X.sum(), (X * X).sum()])}
    for k, seed in ((9, 14), (13, 3)):
        t0 = time.time()
        W0, H0 = _initialize_nmf(X, k, init="nndsvd", random_state=seed)
        print("C3 nndsvd k=%d seed=%d (%.0f s)" % (k, seed, time.time() - t0), flush=True)
        out["k%d_seed" % k] = np.array([seed], dtype=np.int64)
        out["k%d_H0" % k] = H0
        out["k%d_W0_rows" % k] = W0[::97]
        out["k%d_W0_colsum" % k] = np.array([W0.sum(axis=0), (W0 * W0).sum(axis=0)])
    np.savez_compressed(os.path.join(OUT, "ref_c3_nndsvd.npz"), **out)


def make_c4():
    X = c4_matrix()
    out = {"shape": np.array(X.shape), "x_checksum": np.array([float(X.data.astype(np.float64).sum())])}
    X64 = X.astype(np.float64)
    for seed in (11, 12):
        t0 = time.time()
        H, _, n = sklearn_ref.nmf(X64, 20, seed, max_iter=10)
        print("C4 seed=%d: n_iter=%d (%.0f s)" % (seed, n, time.time() - t0), flush=True)
        out["seed%d_H10" % seed] = H.astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "ref_c4_csr.npz"), **out)


def c4_counts_matrix():
    """A COUNT-valued C4: the gamma-Poisson counts of synth.CONFIGS["C4"] (200 000 x 2000, K_true = 20) scaled to unit
    variance per gene like the reference's prepare (cnmf.py:540-548), handed over as CSR float32 -- what a real
    "200k-cell sparse h5ad" is.  The device detects the count structure and takes the f16 integer-plane kernels
    (gemm_mode 4) with 782 cell tiles."""
    X = synth.make_config("C4", dtype=np.float32)
    return sp.csr_matrix(X)


def make_c4_counts():
    X = c4_counts_matrix()
    out = {"shape": np.array(X.shape), "nnz": np.array([X.nnz]),
           "x_checksum": np.array([float(X.data.astype(np.float64).sum())])}
    X64 = X.astype(np.float64)
    for seed in (21, 22):
        t0 = time.time()
        H, _, n = sklearn_ref.nmf(X64, 20, seed, max_iter=10)
        print("C4 counts seed=%d: n_iter=%d (%.0f s)" % (seed, n, time.time() - t0), flush=True)
        out["seed%d_H10" % seed] = H.astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "ref_c4_counts.npz"), **out)


def c4kl_matrix():
    """BASELINE config 4's shape with the library size of a real 10x matrix (e^5.2 -> ~9 % non-zero): the gamma-Poisson
    counts of the C4 topic model (200 000 x 2000, K_true = 20, data seed 3) scaled to unit variance per gene like the
    reference's prepare, handed over as CSR -- the matrix of tools/mu_sparse_probe.py.  (C4KL_CELLS: a smaller dry run.)"""
    C, _ = synth.topic_counts(int(os.environ.get("C4KL_CELLS", 200_000)), 2000, 20, 5.2, 0.4, 3)
    X = synth.normalise_like_prepare(C, dtype=np.float32)
    return sp.csr_matrix(X)


C4KL_JOBS = [(k, seed, T, dt) for T in (100, 20) for (k, seed) in ((20, 31), (9, 32)) for dt in ("float64", "float32")]
C4KL_TMP = os.environ.get("C4KL_TMP", "/tmp/work/c4kl_parts")


def _c4kl_one(args):
    """One scikit-learn run on the CSR matrix; its result goes to its own file at once (a long job: nothing is lost when a
    later one fails)."""
    k, seed, T, dtype = args
    from oracle import nmf_mu_csr
    X = c4kl_matrix().astype(dtype)
    t0 = time.time()
    H, W, n = sklearn_ref.nmf(X, k, seed, beta_loss="kullback-leibler", solver="mu", max_iter=T)
    X64 = X.astype(np.float64)
    ii, jj = X64.nonzero()
    err = nmf_mu_csr.kl_divergence(X64, W.astype(np.float64), H.astype(np.float64), ii, jj)
    print("C4 KL k=%d seed=%d T=%d %s: n_iter=%d err=%.9g (%.0f s)" % (k, seed, T, dtype, n, err, time.time() - t0), flush=True)
    os.makedirs(C4KL_TMP, exist_ok=True)
    np.savez(os.path.join(C4KL_TMP, "k%d_T%d_%s.npz" % (k, T, dtype)), H=H, Whead=W[:4096], Wsum=W.sum(axis=0), n=n, err=err)
    return True


def make_c4kl():
    """Round 5 (review item 1a): scikit-learn ON THE CSR count matrix with beta_loss='kullback-leibler' -- the
    reference's own sparse route (sklearn _nmf.py:192 `_special_sparse_dot`, :526-728 via cnmf.py:672) -- at the size the
    non-zero kernels were benchmarked at: K = 20 and K = 9, 100 and 20 iterations (the stopping rule left on), float64;
    and the same in float32 as the calibration of what single precision delivers there.  ~45 s per iteration and job."""
    import multiprocessing as mp
    X = c4kl_matrix()
    out = {"shape": np.array(X.shape), "nnz": np.array([X.nnz]),
           "x_checksum": np.array([float(X.data.astype(np.float64).sum())])}
    del X
    todo = [j for j in C4KL_JOBS if not os.path.exists(os.path.join(C4KL_TMP, "k%d_T%d_%s.npz" % (j[0], j[2], j[3])))]
    if todo:
        with mp.get_context("fork").Pool(int(os.environ.get("C4KL_PROCS", 4))) as pool:
            pool.map(_c4kl_one, todo, chunksize=1)
    merge_c4kl(out)


def merge_c4kl(out=None):
    """ref_c4_kl.npz from whatever per-job files exist (float64 entries require their job; float32 calibration optional)."""
    from oracle import nmf_cd
    if out is None:
        X = c4kl_matrix()
        out = {"shape": np.array(X.shape), "nnz": np.array([X.nnz]), "x_checksum": np.array([float(X.data.astype(np.float64).sum())])}
    for (k, seed) in ((20, 31), (9, 32)):
        out["k%d_seed" % k] = np.array([seed])
        for T in (20, 100):
            f64 = os.path.join(C4KL_TMP, "k%d_T%d_float64.npz" % (k, T))
            if not os.path.exists(f64):
                continue
            a = np.load(f64)
            out["k%d_H%d" % (k, T)] = a["H"].astype(np.float32)
            out["k%d_Whead%d" % (k, T)] = a["Whead"].astype(np.float32)
            out["k%d_Wsum%d" % (k, T)] = a["Wsum"]
            n32, err32, dev = -1, np.nan, (np.nan, np.nan)
            f32 = os.path.join(C4KL_TMP, "k%d_T%d_float32.npz" % (k, T))
            if os.path.exists(f32):
                b = np.load(f32)
                n32, err32, dev = int(b["n"]), float(b["err"]), nmf_cd.spectra_error(a["H"], b["H"])
            out["k%d_n%d" % (k, T)] = np.array([int(a["n"]), n32])
            out["k%d_err%d" % (k, T)] = np.array([float(a["err"]), err32])
            out["k%d_f32dev%d" % (k, T)] = np.array(dev)
            print("k=%d T=%d: n_iter %d (f32 %d), err %.9g, sklearn float32 vs float64 %s" % (k, T, int(a["n"]), n32, float(a["err"]), dev), flush=True)
    np.savez_compressed(os.path.join(OUT, "ref_c4_kl.npz"), **out)


def make_c4_stop():
    """Round 5 (review item 7): BASELINE config 4 AS STATED -- K = 20, the restarts of the n_iter = 100 ledger (seed 14) run
    to scikit-learn's stopping rule (tol 1e-4, max_iter 1000).  On this synthetic matrix K = 20 = K_true and every one of
    the 100 restarts stops after 36..55 outer iterations (device: gpurun_out/iters_c4.json); the golden holds the shortest
    (ledger row 15: 36) and the longest (row 75: 55), float64, plus the float32 calibration (spectra distance, iteration
    count).  The matrix is handed to scikit-learn DENSE here (BLAS products: ~1 s per iteration instead of ~10 s through
    scipy's CSR product); the coordinate-descent arithmetic on the CSR matrix differs only in the summation order of
    X.H^T / X^T.W (1e-15 relative in float64)."""
    from oracle import nmf_cd
    X32 = synth.make_config("C4", dtype=np.float32)
    X = X32.astype(np.float64)
    led = sklearn_ref.ledger([20], 100, 14)
    out = {"shape": np.array(X.shape), "x_checksum": np.array([float(X.sum())])}
    for row in (15, 75):
        k, it, seed = led[row]
        t0 = time.time()
        H, W, n = sklearn_ref.nmf(X, k, seed)
        obj = float(((X - W @ H) ** 2).sum())
        H32, _, n32 = sklearn_ref.nmf(X32, k, seed)
        dev = nmf_cd.spectra_error(H, H32)
        print("C4 K=20 ledger row %d seed %d: n_iter %d (float32: %d), objective %.9g, float32 vs float64 %s (%.0f s)"
              % (row, seed, n, n32, obj, dev, time.time() - t0), flush=True)
        out["row%d_seed" % row] = np.array([seed, it, n, n32], dtype=np.int64)
        out["row%d_H" % row] = H.astype(np.float32)
        out["row%d_obj" % row] = np.array([obj])
        out["row%d_f32dev" % row] = np.array(dev)
    np.savez_compressed(os.path.join(OUT, "ref_c4_stop.npz"), **out)


C3PIPE_KS = (7, 9, 11)
C3PIPE_NITER = 8
C3PIPE_TMP = os.environ.get("C3PIPE_TMP", "/tmp/work/c3pipe_parts")
_C3X = {}


def _c3pipe_one(job):
    """One scikit-learn restart of the C3 pipeline golden (its own file: a long job loses nothing when a later one fails)."""
    k, it, seed, dt = job
    path = os.path.join(C3PIPE_TMP, "k%d_it%d_%s.npz" % (k, it, dt))
    if os.path.exists(path):
        return True
    from threadpoolctl import threadpool_limits
    t0 = time.time()
    with threadpool_limits(1):
        H, _, n = sklearn_ref.nmf(_C3X[dt], k, seed)
    np.savez(path, H=H.astype(np.float64), n=n)
    print("C3 pipeline k=%d iter=%d seed=%d %s: n_iter=%d (%.0f s)" % (k, it, seed, dt, n, time.time() - t0), flush=True)
    return True


def _c3pipe_consensus(X, led, dt):
    """merged spectra (iter asc, topic asc: cnmf.py:765-770) -> consensus core at the defaults (density threshold 0.5,
    local neighbourhood 0.30) and the stats branch (no density filter: cnmf.py:884-886, 922-936), per K."""
    from oracle import consensus as oc
    out = {}
    for K in C3PIPE_KS:
        parts = [np.load(os.path.join(C3PIPE_TMP, "k%d_it%d_%s.npz" % (k, it, dt))) for (k, it, _) in led if k == K]
        merged = np.concatenate([p["H"] for p in parts], axis=0)
        t0 = time.time()
        core = oc.consensus_core(merged, X, K, density_threshold=0.5)
        stats = oc.consensus_core(merged, X, K, stats_mode=True)
        W = core["rf_usages"]
        out[K] = dict(n_iter=np.array([int(p["n"]) for p in parts], dtype=np.int32), merged=merged,
                      local_density=core["local_density"], density_filter=core["density_filter"],
                      median_spectra=core["median_spectra"], usage_colsum=np.array([W.sum(axis=0), (W * W).sum(axis=0)]),
                      usage_rows=W[::97].copy(), silhouette=np.array([stats["silhouette"]]),
                      prediction_error=np.array([stats["prediction_error"]]),
                      stats_median_spectra=stats["median_spectra"])
        print("C3 pipeline K=%d %s: iterations %s, kept %d of %d, silhouette %.6f, prediction error %.9g (%.0f s)"
              % (K, dt, out[K]["n_iter"].tolist(), int(core["density_filter"].sum()), merged.shape[0],
                 stats["silhouette"], stats["prediction_error"], time.time() - t0), flush=True)
    return out


def make_c3_pipeline():
    """Round 6 (review item 1): the WHOLE pipeline at the headline shape on the CPU reference path.  The C3 matrix
    (50 000 x 2000), the reference's ledger for components 7, 9, 11 with n_iter = 8 and seed 14 (cnmf.py:593-610), every
    restart by scikit-learn in FLOAT64 to the stopping rule (tol 1e-4, max_iter 1000: the call of cnmf.py:672) -> merged
    spectra -> oracle/consensus.py core (cnmf.py:871-920: density filter, KMeans, medians, usage refit) and stats branch
    (cnmf.py:922-936).  tests/test_gpu_golden_big.py runs the same ledger on the device, the device's merged spectra
    through the device's consensus, and compares the consensus spectra at the reference's bar (sum of squared differences
    < 1e-4, tests/test_reproducibility.py:12, 96-115).  scikit-learn's own float32 pipeline is run beside it as the
    calibration of what the working precision alone moves (`c3pipe32`; optional entries)."""
    import multiprocessing as mp
    from oracle import nmf_cd
    X32 = synth.make_config("C3", dtype=np.float32)
    _C3X["float64"] = X32.astype(np.float64)
    _C3X["float32"] = X32
    X = _C3X["float64"]
    led = sklearn_ref.ledger(list(C3PIPE_KS), C3PIPE_NITER, 14)
    os.makedirs(C3PIPE_TMP, exist_ok=True)
    dts = ["float64"] + (["float32"] if "c3pipe32" in sys.argv else [])
    # longest first: k = 11 restarts run to ~1000 iterations
    jobs = sorted([(k, it, seed, dt) for (k, it, seed) in led for dt in dts], key=lambda j: (-j[0], j[3] == "float32"))
    with mp.get_context("fork").Pool(int(os.environ.get("C3PIPE_PROCS", min(8, os.cpu_count() or 1)))) as pool:
        pool.map(_c3pipe_one, jobs, chunksize=1)
    out = {"x_checksum": np.array([X.sum(), (X * X).sum()]), "shape": np.array(X.shape),
           "ks": np.array(C3PIPE_KS), "n_iter_per_k": np.array([C3PIPE_NITER]),
           "ledger": np.array(led, dtype=np.int64)}
    res = _c3pipe_consensus(X, led, "float64")
    for K, d in res.items():
        for name, v in d.items():
            if name == "merged":
                continue                               # (1.7 MB of per-restart spectra: not committed)
            out["k%d_%s" % (K, name)] = v
    if "float32" in dts:
        res32 = _c3pipe_consensus(X, led, "float32")
        for K in C3PIPE_KS:
            ref, m32 = res[K]["median_spectra"], res32[K]["median_spectra"]
            perm, cos = nmf_cd.match_components(ref, m32)
            m32 = m32[perm]
            drift = np.array([((m32 - ref) ** 2).sum(), np.linalg.norm(m32 - ref) / np.linalg.norm(ref),
                              np.abs(m32 - ref).max() / np.abs(ref).max(), cos.min(),
                              abs(int(res32[K]["density_filter"].sum()) - int(res[K]["density_filter"].sum())),
                              abs(res32[K]["silhouette"][0] - res[K]["silhouette"][0]),
                              abs(res32[K]["prediction_error"][0] / res[K]["prediction_error"][0] - 1.0)])
            out["k%d_f32_drift" % K] = drift
            out["k%d_f32_n_iter" % K] = res32[K]["n_iter"]
            print("C3 pipeline K=%d: sklearn float32 pipeline vs float64: sum sq %.3g, rel fro %.3g, rel max %.3g, min cos "
                  "%.6f, |delta kept| %d, |delta silhouette| %.3g, rel delta prediction error %.3g" % (K, *drift), flush=True)
    np.savez_compressed(os.path.join(OUT, "ref_c3_pipeline.npz"), **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["c3", "c4", "c4counts"]
    if "c3pipe" in which or "c3pipe32" in which:
        make_c3_pipeline()
    if "c4stop" in which:
        make_c4_stop()
    if "c4kl" in which:
        make_c4kl()
    if "c4klmerge" in which:
        merge_c4kl()
    if "c4counts" in which:
        make_c4_counts()
    if "c4" in which:
        make_c4()
    if "c3" in which:
        make_c3()
    if "c3" in which or "c3drift" in which:
        make_c3_drift()
    if "c3" in which or "c3nndsvd" in which:
        make_c3_nndsvd()
