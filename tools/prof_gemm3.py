"""Runs the split-operand GEMM a few times (for rocprofv3 --pmc / --kernel-trace)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cnmf_amd.engine import Engine
eng = Engine(0)
rs = np.random.RandomState(0)
shape = sys.argv[1] if len(sys.argv) > 1 else "A"
KC, K, J, ns = (256, 2048, 50048, 1) if shape == "A" else (256, 50048, 2048, 32)
A = rs.rand(KC, K).astype(np.float32)
B = (rs.rand(J, K) * (rs.rand(J, K) < 0.3)).astype(np.float32)
_, ms = eng.debug_gemm3(A, B, nsplit=ns, reps=int(os.environ.get("REPS", "5")))
print("gemm3 %s: %.3f ms" % (shape, ms))
