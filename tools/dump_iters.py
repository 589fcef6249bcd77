"""Iteration counts per restart of the north-star job (900 restarts, K = 5..13, C3) -> gpurun_out/iters_c3.json:
the input of tools/sim_schedule.py (offline simulation of queue orders and batch widths)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cnmf_amd import synth
from cnmf_amd.cnmf import ledger_seeds
from cnmf_amd.engine import Engine

# WORKLOAD=C4 KMIN=20 KMAX=20 N_ITER=100: BASELINE config 4 (-> gpurun_out/iters_c4.json; round 5: which K = 20 restarts are long)
WL = os.environ.get("WORKLOAD", "C3")
X = synth.make_config(WL, dtype=np.float32)
led = ledger_seeds(list(range(int(os.environ.get("KMIN", 5)), int(os.environ.get("KMAX", 13)) + 1)), int(os.environ.get("N_ITER", 200)), 14)
with Engine(0) as eng:
    eng.set_matrix(X)
    t0 = time.time()
    _, _, n_iter, _ = eng.nmf_batch([k for k, _, _ in led], seeds=[int(s) for _, _, s in led], warn=False)
    dt = time.time() - t0
os.makedirs("gpurun_out", exist_ok=True)
json.dump({"k": [int(k) for k, _, _ in led], "iter": [int(i) for _, i, _ in led], "seed": [int(s) for _, _, s in led], "n_iter": [int(n) for n in n_iter],
           "seconds": dt, "stats": {k: (float(v) if not isinstance(v, int) else v) for k, v in eng.last_stats.items()}},
          open("gpurun_out/iters_%s.json" % WL.lower(), "w"))
by = {}
for (k, _, _), n in zip(led, n_iter):
    by.setdefault(k, []).append(int(n))
for k in sorted(by):
    v = np.array(by[k]); print("k=%2d n=%3d mean %6.1f median %6.1f min %4d max %4d p90 %6.1f" % (k, len(v), v.mean(), np.median(v), v.min(), v.max(), np.percentile(v, 90)))
print("%.1f restarts/s" % (len(led) / dt))
