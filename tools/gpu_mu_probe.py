import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cnmf_amd import synth
from cnmf_amd.engine import Engine
X = synth.make_config("C3", dtype=np.float32)
eng = Engine(0); eng.set_matrix(X)
for k in (5, 9, 13, 20):
    t = time.perf_counter()
    H, _, n, err = eng.nmf_mu_batch([k], seeds=[7], max_iter=100, tol=0, warn=False)
    dt = time.perf_counter() - t
    print("KL k=%d: %d iterations in %.3f s -> %.1f us/iteration" % (k, n[0], dt, dt / n[0] * 1e6), flush=True)
