#!/usr/bin/env python
"""bench.py -- NMF restarts/s of the MI355X-native cNMF hot path (BASELINE.json metric).

Workload (default, ``config.workload``): the north-star headline shape -- 50 000 cells x
2000 high-variance genes (synthetic gamma-Poisson counts, reference `prepare` scaling,
cnmf_amd/synth.py "C3"), K in {5..13}.  One STEP = one pass of the hot path over one
batch of restarts: ``--restarts-per-k`` restarts for every K (default 100 = the north star's
n_iter -> one step is one whole factorize() job of 900 restarts per GPU, streamed through 256
packed component columns by the slot work-queue), run to sklearn's stopping rule (tol 1e-4, max_iter 1000)
with sklearn's init='random' generated on the device from the cNMF ledger seeds
(master seed 14).  X is resident in HBM before the timed region.

Multi-GPU (weak scaling, one process per GPU via torch.distributed.run): every rank holds
a replica of X and runs its own batch of restarts per step (ledger rows sharded
round-robin like the reference's worker_filter, cnmf.py:52-53); the only exchange is one
all-gather of the per-restart spectra at the end of the step (RCCL over xGMI).

Prints ONE JSON line on rank 0 (contract in the task statement), with two extra objects:
  roofline     -- the dominant kernel (MFMA GEMM passes): algorithmic flops / launch duration measured
                  with HIP events inside the library (every --event-stride-th iteration, default 64), vs the roofline of the
                  f32-accurate split-operand scheme (bf16 dense MFMA peak / 6)
  cpu_baseline -- scikit-learn's non_negative_factorization (the call the reference makes,
                  cnmf.py:672) timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense
# bf16 MFMAs per f32-accurate product (kernels_gemm3.hip.h), by cnmf_batch_stats.gemm_mode:
#   1/2: both operands as three planes, ah*bh + (ah*bm + am*bh) + (ah*bl + am*bm + al*bh)
#   3  : count-structured X = (integers <= 256) x per-gene scale -> ONE integer plane, (ah + am + al)*n, all exact
#   4  : the same on the f16 pipe (kernels_gemm2h.hip.h): counts <= 2048 in one f16 plane, the factor as TWO f16 planes
#        with a per-row exponent (within 1 ulp_f32 of the f32 value, exact for 3 in 4), partial products exact
SPLIT_MFMAS_PER_PRODUCT = {1: 6, 2: 6, 3: 3, 4: 2}
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="C3", help="C1|C2|C3 (cnmf_amd/synth.py)")
    ap.add_argument("--n-cells", type=int, default=None, help="truncate the workload (debug only)")
    ap.add_argument("--restarts-per-k", type=int, default=100)
    ap.add_argument("--kmin", type=int, default=5)
    ap.add_argument("--kmax", type=int, default=13)
    ap.add_argument("--event-stride", type=int, default=64,
                    help="HIP events around the two GEMM passes of every n-th iteration (the roofline's launch durations)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=30)
    ap.add_argument("--cpu-baseline-child", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-workers-child", default=None, help=argparse.SUPPRESS)
    return ap.parse_args()


def _cpu_worker(job):
    """One cNMF-style worker: a single-threaded scikit-learn restart capped at ``max_iter`` outer iterations."""
    k, seed, max_iter = job
    from threadpoolctl import threadpool_limits
    from oracle import sklearn_ref
    with threadpool_limits(1):
        t0 = time.perf_counter()
        _, _, n_it = sklearn_ref.nmf(_CPU_X64, k, seed=seed, max_iter=max_iter)
        return int(n_it), time.perf_counter() - t0


_CPU_X64 = None


def cpu_workers_child(npy_path, iters):
    """Mode 2 of SURVEY 8d, in its OWN interpreter started with OPENBLAS/OMP/MKL_NUM_THREADS=1 (set before numpy is
    imported, so that every forked worker is single-threaded by construction): one process per available core, each
    running ONE scikit-learn restart capped at ``iters`` outer iterations -- how cNMF is parallelised in practice
    (Extras/run_parallel.py: total_workers independent processes)."""
    global _CPU_X64
    import multiprocessing as mp
    _CPU_X64 = np.load(npy_path).astype(np.float64)
    ks = (5, 9, 13)
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    try:                                                  # cgroup v2 CPU quota, if any
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            ncpu = max(1, min(ncpu, int(float(q) / float(per))))
    except Exception:
        pass
    try:
        avail = [int(l.split()[1]) * 1024 for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0]
    except Exception:
        avail = 64 << 30
    per_worker = int(1.3 * _CPU_X64.nbytes)              # each worker holds its own X^T copy (sklearn _nmf.py:491)
    workers = max(1, min(ncpu, int(0.5 * avail // per_worker)))
    jobs = [(ks[i % len(ks)], 2000 + i, iters) for i in range(workers)]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(workers) as pool:
        res = pool.map(_cpu_worker, jobs, chunksize=1)
    dt = time.perf_counter() - t0
    it = sum(r[0] for r in res)
    busy = max(r[1] for r in res)                         # the slowest worker's own clock: excludes fork/teardown
    print(json.dumps({"restart_iterations_per_s": it / busy, "workers": workers, "iterations": it, "seconds": busy,
                      "wall_seconds_incl_fork": dt, "iterations_per_worker": iters}))


def cpu_baseline_child(npy_path, max_iter):
    """Runs in a FRESH interpreter (no HIP context).  SURVEY.md section 8d: time (1) one worker with all BLAS threads
    here and (2) cNMF-style single-thread worker processes (cpu_workers_child, its own interpreter) on a stratified k
    sample; report both."""
    import subprocess
    from oracle import sklearn_ref
    X64 = np.load(npy_path).astype(np.float64)
    ks = (5, 9, 13)
    try:
        from threadpoolctl import threadpool_info
        info = threadpool_info()
        blas_threads = max([p.get("num_threads", 1) for p in info] + [1])
        blas = ", ".join(sorted({"%s %s" % (p.get("internal_api"), p.get("version")) for p in info}))
    except Exception:
        blas_threads, blas = os.cpu_count(), "unknown"
    # (1) one worker, all BLAS threads
    t0 = time.perf_counter()
    it1 = 0
    for k in ks:
        _, _, n_it = sklearn_ref.nmf(X64, k, seed=1000 + k, max_iter=max_iter)
        it1 += n_it
    dt1 = time.perf_counter() - t0
    del X64
    # (2) single-thread workers, bounded: 3 outer iterations each, 120 s at most
    env = dict(os.environ, OPENBLAS_NUM_THREADS="1", OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
    mode2 = None
    try:
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-workers-child", npy_path, "--cpu-iters", "3"],
                           capture_output=True, text=True, timeout=120, env=env)
        if p.returncode == 0:
            mode2 = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
        else:
            mode2 = {"error": p.stderr[-400:]}
    except subprocess.TimeoutExpired:
        mode2 = {"error": "timed out after 120 s"}
    cpu_model = "unknown"
    try:
        cpu_model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    print(json.dumps({
        "one_worker_all_threads": {"restart_iterations_per_s": it1 / dt1, "threads": int(blas_threads),
                                   "iterations": it1, "seconds": dt1},
        "workers_single_thread": mode2,
        "cpu_count": os.cpu_count(), "cpu_model": cpu_model, "blas": blas, "ks": list(ks)}))


def cpu_baseline(X32, mean_iters_per_restart, max_iter):
    """scikit-learn CD-NMF (float64, the reference dtype, cnmf.py:534) on the host cores -- the call the
    reference makes (cnmf.py:672), in the better of SURVEY 8d's two modes.  The primary number is restart-
    ITERATIONS per second (measured, no extrapolation); restarts/s divides it by the GPU run's mean
    iteration count per restart."""
    import subprocess
    import tempfile
    import shutil
    tmpdir = tempfile.gettempdir()
    try:
        if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 2 * X32.nbytes:
            tmpdir = "/dev/shm"
    except OSError:
        pass
    path = os.path.join(tmpdir, "cnmf_bench_X_%d.npy" % os.getpid())
    np.save(path, X32)
    try:
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", path,
                            "--cpu-iters", str(max_iter)], capture_output=True, text=True, timeout=900)
    finally:
        os.remove(path)
    if p.returncode != 0:
        raise RuntimeError("cpu baseline child failed: %s" % p.stderr[-2000:])
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    a, b = d["one_worker_all_threads"], d["workers_single_thread"]
    ok2 = isinstance(b, dict) and "restart_iterations_per_s" in b
    best_is_workers = ok2 and b["restart_iterations_per_s"] >= a["restart_iterations_per_s"]
    best = b if best_is_workers else a
    it_per_s = best["restart_iterations_per_s"]
    return {
        "value": it_per_s,
        "unit": "restart-iterations/s",
        "cores": int(b["workers"] if best_is_workers else a["threads"]),
        "kind": "reference",
        "mode": ("%d single-thread worker processes (cNMF's own parallelism, total_workers = workers)" % b["workers"]
                 if best_is_workers else "1 worker x %d BLAS threads" % a["threads"]),
        "restarts_per_s_extrapolated": it_per_s / max(mean_iters_per_restart, 1.0),
        "modes": d,
        "sample": ("sklearn.decomposition.non_negative_factorization (solver=cd, float64, init=random) on the same X, "
                   "k in (5, 9, 13): mode 1 = one worker with all BLAS threads, %d outer iterations per k (%d iterations "
                   "in %.1f s); mode 2 = one single-thread process per core, one restart each capped at 3 outer iterations "
                   "(%s); the better mode is `value`; restarts_per_s_extrapolated = value / mean iterations per restart of "
                   "the GPU run (%.1f)"
                   % (max_iter, a["iterations"], a["seconds"],
                      ("%d workers, %d iterations, slowest worker %.1f s" % (b["workers"], b["iterations"], b["seconds"]))
                      if ok2 else "failed: %s" % (b or {}).get("error", "?"), mean_iters_per_restart)),
    }


def pmc_traffic(key, split_operand):
    """HBM bytes per launch of the dominant kernel from the PMC counters (FETCH_SIZE, WRITE_SIZE),
    collected in separate rocprofv3 --pmc passes (tools/gpu_pmc_bench.sh) and committed under
    profiles/ -- counters cannot be read from inside an un-profiled run.  None if absent."""
    name = {0: "r1_pmc_traffic.json", 1: "r1_pmc_traffic_split.json", 2: "r1_pmc_traffic_split.json",
            3: "r1_pmc_traffic_counts.json", 4: "r2_pmc_traffic_f16.json"}[split_operand]
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
        return {"hbm_bytes_per_launch": d[key]["hbm_bytes_per_launch"],
                "algorithmic_bytes_per_launch": d["algorithmic_bytes_per_launch"][key],
                "source": "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, FETCH doubled per the gfx950 note)" % name}
    except Exception:
        return None


def consensus_wallclock(eng, with_cpu=True):
    """BASELINE config 5 (consensus-only stress: 5000 stacked spectra x 2000 genes, k=20):
    wall-clock of the consensus core on the GPU (host call incl. transfers) next to the
    reference's sklearn/pandas calls on the host cores.  Reported beside the headline metric."""
    from cnmf_amd import synth
    S, _ = synth.consensus_stress(R=5000, G=2000, k=20, n_outliers=100, seed=0)
    eng.consensus(S, 20, density_threshold=0.5)                     # warm-up (allocations, code objects)
    t0 = time.perf_counter()
    out = eng.consensus(S, 20, density_threshold=0.5)
    gpu_s = time.perf_counter() - t0
    res = {"workload": "C5: 5000 spectra x 2000 genes, k=20, density_threshold 0.5, n_neighbors 75",
           "gpu_ms": 1e3 * gpu_s, "rows_kept": int(out["n_kept"]), "dtype": "f64"}
    if with_cpu:
        import pandas as pd
        from sklearn.cluster import KMeans
        from sklearn.metrics.pairwise import euclidean_distances
        t0 = time.perf_counter()
        l2 = (S.T / np.sqrt((S ** 2).sum(axis=1))).T
        D = euclidean_distances(l2)
        n = int(0.30 * S.shape[0] / 20)
        po = np.argpartition(D, n + 1)[:, :n + 1]
        dens = D[np.arange(D.shape[0])[:, None], po].sum(1) / n
        l2k = l2[dens < 0.5]
        km = KMeans(n_clusters=20, n_init=10, random_state=1).fit(l2k)
        pd.DataFrame(l2k).groupby(pd.Series(km.labels_ + 1)).median()
        res["cpu_reference_ms"] = 1e3 * (time.perf_counter() - t0)
        res["cpu_cores"] = os.cpu_count()
        res["labels_match_cpu"] = bool(np.array_equal(out["labels"][out["density_filter"]], km.labels_))
    return res


def main():
    args = parse()
    if args.cpu_baseline_child:
        cpu_baseline_child(args.cpu_baseline_child, args.cpu_iters)
        return
    if args.cpu_workers_child:
        cpu_workers_child(args.cpu_workers_child, args.cpu_iters)
        return
    # stdout must carry exactly ONE JSON line, but RCCL prints a version banner to the C-level
    # stdout (flushed at exit): keep the real stdout aside and point fd 1 at stderr meanwhile
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    multi = world > 1 or bool(os.environ.get("CNMF_BENCH_FORCE_DIST"))     # (the override exercises the N > 1 code at world 1)
    if os.environ.get("CNMF_BENCH_ONE_GPU"):                               # test hook: every rank on GPU 0
        local_rank = 0
    # Transport of everything that crosses ranks (the one data-path gather, the barrier, the max/sum over ranks):
    #   "rccl"  (default) -- ncclAllGather inside the C-ABI library (cnmf_comm_* / cnmf_allgather_*); NO torch:
    #                        the launcher only provides RANK / WORLD_SIZE / MASTER_PORT, the 128-byte RCCL id
    #                        travels through a file;
    #   "torch" -- torch.distributed (backend nccl = RCCL, or gloo for the one-GPU test hook).
    gather_mode = "none"
    if multi:
        gather_mode = os.environ.get("CNMF_GATHER", "torch" if os.environ.get("CNMF_BENCH_BACKEND") else "rccl")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # before the HIP runtime starts (dmabuf IPC only)
    dist = None
    if gather_mode == "torch":
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch
        import torch.distributed as dist
        # (test hook: CNMF_BENCH_BACKEND=gloo CNMF_BENCH_ONE_GPU=1 runs several ranks on ONE GPU to exercise the
        #  N > 1 bookkeeping -- ledger sharding, ragged gather, max-over-ranks -- where only one GPU exists)
        backend = os.environ.get("CNMF_BENCH_BACKEND", "nccl")
        if os.environ.get("CNMF_BENCH_ONE_GPU"):
            local_rank = 0
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend)
    tdev = "cpu" if os.environ.get("CNMF_BENCH_BACKEND", "nccl") == "gloo" else "cuda"

    from cnmf_amd import synth
    from cnmf_amd.cnmf import ledger_seeds
    from cnmf_amd.engine import Engine

    X = synth.make_config(args.workload, dtype=np.float32, n_cells=args.n_cells)
    N, G = X.shape
    eng = Engine(local_rank)
    eng.set_matrix(X)
    if gather_mode == "rccl":
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        from cnmf_amd import dist as cd
        # all ranks of one node are children of the same launcher process: its pid + the rendezvous port
        # name the id file uniquely for this launch
        id_path = os.environ.get("CNMF_RCCL_ID_FILE") or os.path.join(
            "/tmp", "cnmf_rccl_id.%d.%s" % (os.getppid(), os.environ.get("MASTER_PORT", "0")))
        cd.comm_bootstrap_file(eng, rank, world, id_path)

    ks_all = list(range(args.kmin, args.kmax + 1))
    n_steps_total = args.warmup + args.steps
    led = ledger_seeds(ks_all, args.restarts_per_k * n_steps_total * world, 14)      # the product's own ledger
    by_k = {k: [s for (kk, _, s) in led if kk == k] for k in ks_all}

    def step_jobs(step):
        ks, seeds = [], []
        for k in ks_all:
            base = (step * world + rank) * args.restarts_per_k
            for j in range(args.restarts_per_k):
                ks.append(k)
                seeds.append(by_k[k][base + j])
        return ks, seeds

    def barrier():
        # every engine call returns with its stream drained, so a barrier over ranks is also a device barrier
        if gather_mode == "rccl":
            eng.allgather_array(np.zeros(1, dtype=np.int64))     # one tiny ncclAllGather + stream sync
        elif dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    def reduce_over_ranks(elapsed, restarts, riters):
        """MAX of the elapsed time, SUM of the counters."""
        if gather_mode == "rccl":
            v = eng.allgather_array(np.array([elapsed, restarts, riters], dtype=np.float64))
            return float(v[:, 0].max()), float(v[:, 1].sum()), float(v[:, 2].sum())
        if dist is not None:
            import torch
            t = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            cnt = torch.tensor([restarts, riters], dtype=torch.float64, device=tdev)
            dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
            return float(t.item()), float(cnt[0].item()), float(cnt[1].item())
        return elapsed, float(restarts), float(riters)

    def gather(H_list, ks=None, step=0):
        """The one data-path collective: all-gather of the packed per-restart spectra."""
        if gather_mode == "none":
            return
        from cnmf_amd import dist as cd
        if gather_mode == "rccl":
            hdr = np.array([(i, int(k), rank * len(ks) + i) for i, k in enumerate(ks)], dtype=np.int32).reshape(-1, 3)
            merged = cd.allgather_spectra_rccl(eng, hdr, None, G)      # blk=None: the device-resident spectra store
            assert len(merged) == world * len(ks), (len(merged), world, len(ks))
            return
        import torch
        rows = [(i, int(H.shape[0]), step) for i, H in enumerate(H_list)]
        hdr, blk = cd.pack_local(rows, H_list, G)
        cd.allgather_spectra(hdr, blk, G, device=None if tdev == "cpu" else "cuda:%d" % local_rank)
        torch.cuda.synchronize()

    def run_step(step, profile):
        ks, seeds = step_jobs(step)
        if gather_mode == "rccl":
            eng.spectra_reset()
            _, _, n_iter, _ = eng.nmf_batch(ks, seeds=seeds, warn=False, profile=args.event_stride if profile else 0, resident=True)
            st = dict(eng.last_stats)
            gather(None, ks, step)
        else:
            H, _, n_iter, _ = eng.nmf_batch(ks, seeds=seeds, warn=False, profile=args.event_stride if profile else 0)
            st = dict(eng.last_stats)
            gather(H, ks, step)
        return ks, st

    agg = dict(restarts=0, restart_iters=0, rc_iters=0, outer=0, col_iters=0, passA_ms=0.0,
               passB_ms=0.0, nA=0, nB=0, gpu_ms=0.0, kc=0, nsplit=0, gemm_mode=0)
    for step in range(args.warmup):
        run_step(step, False)
    barrier()
    t0 = time.perf_counter()
    for step in range(args.warmup, n_steps_total):
        ks, st = run_step(step, True)
        agg["restarts"] += len(ks)
        agg["restart_iters"] += int(st["restart_iterations"])
        agg["rc_iters"] += int(st["restart_column_iterations"])
        agg["outer"] += int(st["outer_iterations"])
        agg["col_iters"] += int(st["column_iterations"])
        agg["passA_ms"] += st["passA_ms"]; agg["passB_ms"] += st["passB_ms"]
        agg["nA"] += int(st["passA_launches"]); agg["nB"] += int(st["passB_launches"])
        agg["gpu_ms"] += st["gpu_ms"]; agg["kc"] = int(st["kc"]); agg["nsplit"] = int(st["nsplit"])
        agg["gemm_mode"] = int(st["gemm_mode"])
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed, total_restarts, total_riters = reduce_over_ranks(elapsed, agg["restarts"], agg["restart_iters"])

    if rank == 0:
        # roofline of the dominant kernel (rank 0's launches): the MFMA GEMM pass
        flops_per_col_iter = 2.0 * N * G                        # one pass, one component column
        alg_flops_A = flops_per_col_iter * agg["rc_iters"]      # algorithmic (converged columns excluded)
        # the passes of every n-th iteration (--event-stride) are bracketed by HIP events (an event record costs ~6 us of
        # queue time): average duration of the sampled launches x all launches = time in the pass
        launches = max(agg["outer"], 1)
        avgA = agg["passA_ms"] / max(agg["nA"], 1)
        avgB = agg["passB_ms"] / max(agg["nB"], 1)
        tfA = alg_flops_A / max(avgA * launches, 1e-9) / 1e9
        tfB = alg_flops_A / max(avgB * launches, 1e-9) / 1e9
        dom = "A" if avgA >= avgB else "B"
        ach = tfA if dom == "A" else tfB
        split = agg["gemm_mode"] > 0
        if split:
            # f32-accurate products on the bf16 matrix pipe: 6 (3 for count-structured X) bf16 MFMAs per product,
            # so the roofline of the scheme in f32-equivalent flops is the dense bf16 peak / 6 (/ 3)
            per_product = SPLIT_MFMAS_PER_PRODUCT[agg["gemm_mode"]]
            peak = BF16_MFMA_PEAK_TFLOPS / per_product
            if agg["gemm_mode"] == 4:
                kern = ("gemm2h_streamk_kernel (pass A: X.Ht; X = one integer f16 plane x per-gene scale, factor = 2 f16 planes)"
                        if dom == "A" else
                        "gemm2h_kernel (pass B: Xt.W; X = one integer f16 plane x per-gene scale, factor = 2 f16 planes)")
            elif agg["gemm_mode"] == 3:
                kern = ("gemm3c_streamk_kernel (pass A: X.Ht; X = one integer bf16 plane x per-gene scale, factor = 3 planes)"
                        if dom == "A" else
                        "gemm3c_kernel (pass B: Xt.W; X = one integer bf16 plane x per-gene scale, factor = 3 planes)")
            else:
                kern = ("gemm3g_streamk_kernel (pass A: X.Ht, 3 x bf16 planes)" if dom == "A"
                        else "gemm3g_kernel (pass B: Xt.W, 3 x bf16 planes)")
        else:
            peak = FP32_MFMA_PEAK_TFLOPS
            kern = "gemm_streamk_kernel<NT> (pass A: X.Ht)" if dom == "A" else "gemm_kernel<NN> (pass B: Xt.W)"
        xbytes = {0: 4, 1: 6, 2: 6, 3: 2, 4: 2}[agg["gemm_mode"]]     # bytes per element of X as the GEMM reads it
        roof = {
            "bound": "mfma",
            "kernel": kern,
            "achieved": ach, "peak": peak, "unit": "TFLOP/s",
            "flop_basis": ("f32-equivalent flops (2.N.G per column and pass); peak = bf16/f16 dense MFMA peak / %d MFMAs per product"
                           % SPLIT_MFMAS_PER_PRODUCT[agg["gemm_mode"]] if split else "f32 flops; peak = f32 MFMA peak"),
            "frac": ach / peak,
            "traffic": (pmc_traffic("passA" if dom == "A" else "passB", agg["gemm_mode"]) or {}).get("hbm_bytes_per_launch"),
            "traffic_detail": pmc_traffic("passA" if dom == "A" else "passB", agg["gemm_mode"]),
            "avg_launch_ms": {"passA": avgA, "passB": avgB},
            "launches": {"per_pass": agg["outer"], "timed_with_hip_events": agg["nA"]},
            "achieved_passA": tfA, "achieved_passB": tfB,
            "alg_flops_per_launch": alg_flops_A / launches,
            "issued_flops_per_launch": flops_per_col_iter * agg["kc"],
            "x_stream_GBs": {"passA": N * G * xbytes / max(avgA, 1e-9) / 1e6,
                             "passB": N * G * xbytes / max(avgB, 1e-9) / 1e6,
                             "peak": HBM_PEAK_GBS},
            "gemm_share_of_gpu_time": (avgA + avgB) * launches / max(agg["gpu_ms"], 1e-9),
        }
        tr = roof["traffic_detail"]
        if tr:
            # the pass is nearly balanced between the two roofs: also quote the HBM side (PMC bytes / measured time)
            dur = (avgA if dom == "A" else avgB) * 1e-3
            roof["hbm"] = {"achieved_GBs": tr["hbm_bytes_per_launch"] / dur / 1e9, "peak_GBs": HBM_PEAK_GBS,
                           "frac": tr["hbm_bytes_per_launch"] / dur / 1e9 / HBM_PEAK_GBS,
                           "traffic_over_algorithmic": tr["hbm_bytes_per_launch"] / tr["algorithmic_bytes_per_launch"]}
        if agg["gemm_mode"] == 4:
            # measured ceiling of THIS instruction stream with everything but the MFMAs removed (tools/
            # probe_gemm2h_ablate.py var 7, profiles/r2_probe_gemm2h_variants.txt): the matrix pipe on non-zero data at
            # the clock the power budget allows -- not a roofline, but the reason `frac` cannot approach 1
            roof["mfma_only_ablation"] = {"tflops_issued": 1460.0, "source": "profiles/r2_probe_gemm2h_variants.txt",
                                          "issued_over_mfma_only": ach * per_product * agg["col_iters"] / max(agg["rc_iters"], 1) / 1460.0}
        if split:
            roof["matrix_pipe"] = {
                "scheme": ("X = n * d detected (n integer <= 2048: one exact f16 plane; d per gene, folded into the factor); "
                           "factor = 2 f16 planes of a * 2^s_row (within 1 ulp_f32, exact for 3 values in 4); 2 exact partial "
                           "products per product; f32 accumulate" if agg["gemm_mode"] == 4 else
                           "X = n * d detected (n integer <= 256: one exact bf16 plane; d per gene, folded into the factor); "
                           "factor = 3 bf16 planes; 3 exact partial products per product; f32 accumulate"
                           if agg["gemm_mode"] == 3 else
                           "x = h + m + l (3 bf16 planes); a*b from the 6 partial products of weight >= 2^-18; f32 accumulate"),
                "mfma_tflops_issued": ach * per_product * agg["col_iters"] / max(agg["rc_iters"], 1),
                "bf16_dense_peak": BF16_MFMA_PEAK_TFLOPS,
                "vs_f32_matrix_peak": ach / FP32_MFMA_PEAK_TFLOPS,
                "note": "launch averages include the tail launches (< 256 packed columns) that run on the exact-f32 pipe",
            }
        mean_it = total_riters / max(total_restarts, 1.0)
        out = {
            "metric": "NMF restarts/sec (%dx%dxK%d..%d)" % (N, G, args.kmin, args.kmax),
            "value": total_restarts / elapsed,
            "unit": "restarts/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / max(args.steps, 1),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            # f32 results; the products run on the bf16 matrix pipe from exact operand planes
            "dtype": ({1: "f32 (3x3 bf16 planes, f32 accumulate)", 2: "f32 (3x3 bf16 planes, f32 accumulate)",
                       3: "f32 (exact integer bf16 plane x 3 bf16 planes, f32 accumulate)",
                       4: "f32 (exact integer f16 plane x 2 f16 planes with per-row exponent, f32 accumulate)"}.get(agg["gemm_mode"], "f32")),
            "data": "synthetic",
            "config": {"workload": "%s: %d cells x %d HVGs synthetic dense, K=%d..%d, %d restarts per K per step "
                                   "per GPU, sklearn CD solver tol=1e-4 max_iter=1000, init=random from ledger seeds"
                                   % (args.workload, N, G, args.kmin, args.kmax, args.restarts_per_k),
                       "restarts_per_step_per_gpu": len(ks_all) * args.restarts_per_k,
                       "packed_columns": agg["kc"], "splitk_passB": agg["nsplit"],
                       "mean_iterations_per_restart": mean_it,
                       "restart_iterations_per_s": total_riters / elapsed,
                       "column_utilisation": agg["rc_iters"] / max(agg["col_iters"], 1),
                       "parallelism": "restart-sharded x%d" % world, "gather": gather_mode,
                       "torch_in_process": "torch" in sys.modules},
            "roofline": roof,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(X, mean_it, args.cpu_iters)
            out["consensus"] = consensus_wallclock(eng)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)
    barrier()
    if dist is not None:
        dist.destroy_process_group()
    if gather_mode == "rccl":
        eng.comm_finalize()
    eng.close()


if __name__ == "__main__":
    main()
