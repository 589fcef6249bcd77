"""NNDSVD-initialised restarts at the north-star shape (C3, K = 5..13): time of the batched initialisation
(Engine.nndsvd_init_batch: range finders of up to 13 restarts per pass over X, LU / QR / SVD on a host thread pool), of
the per-restart loop it replaces (a sample), and of the restarts themselves -> gpurun_out/nndsvd_c3.json."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cnmf_amd import synth
from cnmf_amd.cnmf import ledger_seeds
from cnmf_amd.engine import Engine

per_k = int(os.environ.get("PER_K", 10))
X = synth.make_config("C3", dtype=np.float32)
led = ledger_seeds(list(range(5, 14)), per_k, 14)
ks, seeds = [k for k, _, _ in led], [int(s) for _, _, s in led]
with Engine(0) as eng:
    eng.set_matrix(X)
    eng.nndsvd_init_batch(ks[:2], seeds[:2])                      # warm-up
    t0 = time.perf_counter(); inits = eng.nndsvd_init_batch(ks, seeds); t_batch = time.perf_counter() - t0
    t0 = time.perf_counter()
    for k, s in list(zip(ks, seeds))[::9][:5]:
        eng.nndsvd_init(k, s)
    t_loop = (time.perf_counter() - t0) / 5
    t0 = time.perf_counter()
    H, _, n_iter, _ = eng.nmf_batch(ks, W0=[w for w, _ in inits], H0=[h for _, h in inits], warn=False)
    t_run = time.perf_counter() - t0
res = {"config": "C3 50000 x 2000, K=5..13, %d restarts with init='nndsvd'" % len(ks),
       "nndsvd_init_batch_s": t_batch, "per_restart_ms_batched": 1e3 * t_batch / len(ks),
       "per_restart_ms_single_calls": 1e3 * t_loop, "speedup_of_batching": t_loop * len(ks) / t_batch,
       "restarts_s": t_run, "mean_iterations": float(np.mean(n_iter)),
       "restarts_per_s_incl_init": len(ks) / (t_batch + t_run), "restarts_per_s_excl_init": len(ks) / t_run,
       "host_threads": len(os.sched_getaffinity(0))}
print(json.dumps(res, indent=1))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/nndsvd_c3.json", "w"), indent=1)
