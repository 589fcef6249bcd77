"""Timing ablations of the f16 two-plane count GEMM (nsub 2): var 4 = production stream, 2 = without MFMAs (fill +
fragment reads only), 3 = without steady-state DMA (MFMAs + fragment reads + barriers), 6 = MFMAs + barriers (fragments
read once), 7 = MFMAs alone (no reads, no barriers, no DMA)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cnmf_amd.engine import Engine
eng = Engine(0)
rs = np.random.RandomState(0)
for K, J, ns, tag in [(2048, 50176, 1, "passA-shape (196 tiles)"), (50176, 2048, 32, "passB ns32")]:
    A = rs.rand(256, K).astype(np.float32)
    B = (rs.poisson(1.0, size=(J, K))).astype(np.float32)
    fl = 2.0 * 256 * K * J
    for rep in range(2):
        for var in (4, 3, 6, 7):
            _, ms = eng.debug_gemm2h(A, B, nsplit=ns, nsub=2 + 16 * var, reps=20)
            print("%s var=%d: %.4f ms (%.0f TF f16-issued equivalent)" % (tag, var, ms, 2 * fl / ms / 1e9), flush=True)
