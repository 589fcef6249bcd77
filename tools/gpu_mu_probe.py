"""KL multiplicative update at C3 (50000 x 2000): time per restart-iteration, one restart alone and batches
(the matrix-pipe path batches up to 16 restarts per round of launches); CNMF_MU_VALU=1 selects the vector-ALU path."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cnmf_amd import synth
from cnmf_amd.engine import Engine
X = synth.make_config("C3", dtype=np.float32)
eng = Engine(0); eng.set_matrix(X)
its = int(os.environ.get("MU_ITERS", "50"))
eng.nmf_mu_batch([5], seeds=[1], max_iter=3, tol=0, warn=False)          # warm up (X^T copy, code objects)
for ks in ([9], [13], [20], [9] * 16, [9] * 32, [5, 6, 7, 8, 9, 10, 11, 12, 13] * 4, [9] * 64, [20] * 16, [20] * 32):
    t = time.perf_counter()
    H, _, n, err = eng.nmf_mu_batch(ks, seeds=list(range(7, 7 + len(ks))), max_iter=its, tol=0, warn=False)
    dt = time.perf_counter() - t
    print("KL k=%s x%d: %d iterations each in %.3f s -> %.1f us per restart-iteration"
          % (sorted(set(ks)), len(ks), n[0], dt, dt / n.sum() * 1e6), flush=True)
if not os.environ.get("CNMF_MU_VALU"):
    eng.set_matrix(X + np.float32(1e-3))                   # Itakura-Saito wants a positive matrix
    ks = [9] * 16
    t = time.perf_counter()
    H, _, n, err = eng.nmf_mu_batch(ks, seeds=list(range(7, 7 + len(ks))), beta_loss="itakura-saito", max_iter=its, tol=0, warn=False)
    dt = time.perf_counter() - t
    print("IS k=[9] x16: %d iterations each in %.3f s -> %.1f us per restart-iteration" % (n[0], dt, dt / n.sum() * 1e6), flush=True)
