"""KL multiplicative update at C3, batches of 16 only (A/B runs of kernel variants)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cnmf_amd import synth
from cnmf_amd.engine import Engine
X = synth.make_config("C3", dtype=np.float32)
eng = Engine(0); eng.set_matrix(X)
its = int(os.environ.get("MU_ITERS", "100"))
eng.nmf_mu_batch([5] * 16, seeds=list(range(16)), max_iter=3, tol=0, warn=False)
for ks in ([9] * 16, [20] * 16):
    for rep in range(2):
        t = time.perf_counter()
        H, _, n, err = eng.nmf_mu_batch(ks, seeds=list(range(7, 7 + len(ks))), max_iter=its, tol=0, warn=False)
        dt = time.perf_counter() - t
        print("KL k=%d x%d: %.1f us per restart-iteration  (err %.6f)" % (ks[0], len(ks), dt / n.sum() * 1e6, err[0]), flush=True)
