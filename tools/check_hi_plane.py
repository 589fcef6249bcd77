"""Which path is closer to the float64 oracle on a count matrix with one 40000-count outlier?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cnmf_amd import synth
from cnmf_amd.engine import Engine
from oracle import nmf_cd
C, _ = synth.topic_counts(1024, 520, 6, 5.0, 0.3, 2)
C = C[:, C.sum(axis=0) > 0]; C = C[C.sum(axis=1) > 0]
X = (C / C.std(axis=0, ddof=1)).astype(np.float64)
for (i, g, c) in ((3, 5, 300), (700, 400, 1000), (11, 17, 40000)):
    X[i, g] = X[:, g][X[:, g] > 0].min() * c / C[:, g][C[:, g] > 0].min()
eng = Engine(0)
ks, seeds = [9] * 29, list(range(1, 30))
res = {}
for mode in ("3", "2", "0"):
    os.environ["CNMF_GEMM3"] = mode
    eng.set_matrix(X)
    H, _, n, _ = eng.nmf_batch(ks, seeds=seeds, max_iter=20, warn=False)
    res[mode] = H
    print("mode", mode, "gemm_mode", eng.last_stats["gemm_mode"])
for r in (0, 7, 20):
    _, H_ref, _ = nmf_cd.nmf(X, 9, seed=seeds[r], max_iter=20)
    for mode in ("3", "2", "0"):
        d = np.abs(res[mode][r] - H_ref).max() / np.abs(H_ref).max()
        print("restart", r, "mode", mode, "max|H - H_f64| / max|H_f64| = %.3e" % d)
