"""Throughput of the restart engine on the small BASELINE configs (C1: ~1000 x 500, K=7, 20 restarts; C2: 2700 x 2000,
K=10, 100 restarts): these are latency / launch bound, not bandwidth bound."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cnmf_amd import synth
from cnmf_amd.cnmf import ledger_seeds
from cnmf_amd.engine import Engine
for cfg, k, n in (("C1", 7, 20), ("C2", 10, 100)):
    X = synth.make_config(cfg, dtype=np.float32)
    eng = Engine(0); eng.set_matrix(X)
    seeds = [s for (_, _, s) in ledger_seeds([k], n, 14)]
    eng.nmf_batch([k] * 4, seeds=seeds[:4], warn=False)
    for rep in range(2):
        t = time.perf_counter()
        H, _, n_iter, _ = eng.nmf_batch([k] * n, seeds=seeds, warn=False)
        dt = time.perf_counter() - t
        st = eng.last_stats
        print("%s %s k=%d x%d: %.3f s -> %.0f restarts/s, %d outer iterations (%.1f us each), gpu_ms %.1f, mean n_iter %.0f, kc %d"
              % (cfg, X.shape, k, n, dt, n / dt, st["outer_iterations"], dt / max(1, st["outer_iterations"]) * 1e6, st["gpu_ms"], np.mean(n_iter), st["kc"]), flush=True)
