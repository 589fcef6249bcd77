"""GPU probe for rocprofv3 --pmc: run ONE GEMM configuration a few times.
usage: prof_gemm.py MODE VARIANT KC NSPLIT [N]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cnmf_amd.engine import Engine  # noqa: E402

mode, variant, KC, ns = [int(x) for x in sys.argv[1:5]]
N = int(sys.argv[5]) if len(sys.argv) > 5 else 50048
G = 2016
eng = Engine(0)
rs = np.random.RandomState(0)
X = np.abs(rs.standard_normal((N, G))).astype(np.float32)
A = np.abs(rs.standard_normal((KC, G if mode == 0 else N))).astype(np.float32)
_, ms = eng.debug_gemm(mode, A, X, variant=variant, nsplit=ns, reps=int(os.environ.get("REPS", 10)))
print("mode %d variant %d KC %d nsplit %d: %.3f ms  %.1f TF" % (mode, variant, KC, ns, ms, 2.0 * N * G * KC / ms / 1e9))
