"""GPU probe: pass A / pass B variants at KC=128/256 (north-star shape)."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cnmf_amd.engine import Engine
N = int(os.environ.get("PROBE_N", 50048)); G = 2016
eng = Engine(0)
rs = np.random.RandomState(0)
X = np.abs(rs.standard_normal((N, G))).astype(np.float32)
for KC in (128, 256):
    H = np.abs(rs.standard_normal((KC, G))).astype(np.float32)
    Wt = np.abs(rs.standard_normal((KC, N))).astype(np.float32)
    for variant in (1, 2, 3):
        _, ms = eng.debug_gemm(0, H, X, variant=variant, reps=10)
        print("A KC=%d v%d: %.3f ms %.1f TF" % (KC, variant, ms, 2.0*N*G*KC/ms/1e9), flush=True)
    for variant in (1, 2, 3):
        for ns in (8, 16, 32):
            _, ms = eng.debug_gemm(1, Wt, X, variant=variant, nsplit=ns, reps=10)
            print("B KC=%d v%d ns=%d: %.3f ms %.1f TF" % (KC, variant, ns, ms, 2.0*N*G*KC/ms/1e9), flush=True)
