"""Do two factorize pipelines on ONE GPU overlap?  (two contexts, two host threads, two HIP streams:
the HBM-bound sweeps / splits of one batch under the MFMA-bound GEMM passes of the other.)
  a) one engine, the C3 north-star ledger (900 restarts);  b) two engines, half the ledger each, concurrently."""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cnmf_amd import synth
from cnmf_amd.cnmf import ledger_seeds
from cnmf_amd.engine import Engine

per_k = int(os.environ.get("PER_K", "100"))
X = synth.make_config("C3", dtype=np.float32)
ks_all = list(range(5, 14))
led = ledger_seeds(ks_all, per_k * 2, 14)
by_k = {k: [s for (kk, _, s) in led if kk == k] for k in ks_all}

def jobs(lo, hi):
    ks, seeds = [], []
    for k in ks_all:
        for j in range(lo, hi):
            ks.append(k); seeds.append(by_k[k][j])
    return ks, seeds

engs = [Engine(0), Engine(0)]
for e in engs:
    e.set_matrix(X)
    e.nmf_batch(*jobs(0, 3)[:1], seeds=jobs(0, 3)[1], warn=False, resident=True)      # warm up (planes, buffers)

def run(e, lo, hi, out, i):
    ks, seeds = jobs(lo, hi)
    e.spectra_reset()
    _, _, n_iter, _ = e.nmf_batch(ks, seeds=seeds, warn=False, resident=True)
    out[i] = (len(ks), int(np.sum(n_iter)), dict(e.last_stats))

MODE = os.environ.get("MODE", "both")
for rep in range(int(os.environ.get("REPS", "2"))):
  if MODE in ("both", "one"):
    out = [None, None]
    t = time.perf_counter(); run(engs[0], per_k, 2 * per_k, out, 0); dt1 = time.perf_counter() - t
    n1, it1, st1 = out[0]
    print("one engine : %d restarts %.3f s -> %.1f restarts/s  (%d restart-iterations, %d outer its, passA %.1f us passB %.1f us)"
          % (n1, dt1, n1 / dt1, it1, st1["outer_iterations"], 1e3 * st1["passA_ms"] / max(1, st1["passA_launches"]),
             1e3 * st1["passB_ms"] / max(1, st1["passB_launches"])), flush=True)
  if MODE in ("both", "two"):
    out = [None, None]
    half = per_k + per_k // 2
    th = [threading.Thread(target=run, args=(engs[0], per_k, half, out, 0)),
          threading.Thread(target=run, args=(engs[1], half, 2 * per_k, out, 1))]
    t = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    dt2 = time.perf_counter() - t
    n2 = out[0][0] + out[1][0]
    print("two engines: %d restarts %.3f s -> %.1f restarts/s  (%d restart-iterations; outer its %d + %d)"
          % (n2, dt2, n2 / dt2, out[0][1] + out[1][1], out[0][2]["outer_iterations"], out[1][2]["outer_iterations"]), flush=True)
