"""Timing ablations of the f16 two-plane count GEMM (nsub 2): var 4 = production stream, 2 = without MFMAs (fill +
fragment reads only), 3 = without steady-state DMA (MFMAs + fragment reads + barriers), 6 = MFMAs + barriers (fragments
read once), 7 = MFMAs alone (no reads, no barriers, no DMA)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cnmf_amd.engine import Engine
eng = Engine(0)
rs = np.random.RandomState(0)
# round 5: the production pass-B launch itself (1024 packed columns x 8 K splits = 256 workgroups of 196 steps each) beside the
# 256-column probe shapes of rounds 2-4 (49 steps per workgroup: there the tile store + prologue are a quarter of the launch)
VARS = tuple(int(v) for v in os.environ.get("ABLATE_VARS", "4,3,6,7").split(","))
shapes = [(256, 2048, 50176, 1, "passA-shape (196 tiles)"), (256, 50176, 2048, 32, "passB ns32 (49 steps per workgroup)"),
          (1024, 50176, 2048, 8, "passB production: 1024 columns, ns8 (196 steps per workgroup)")]
if os.environ.get("ABLATE_SHAPES"):
    shapes = [shapes[int(i)] for i in os.environ["ABLATE_SHAPES"].split(",")]
for KC, K, J, ns, tag in shapes:
    A = rs.rand(KC, K).astype(np.float32)
    B = (rs.poisson(1.0, size=(J, K))).astype(np.float32)
    fl = 2.0 * KC * K * J
    for rep in range(2):
        for var in VARS:
            _, ms = eng.debug_gemm2h(A, B, nsplit=ns, nsub=2 + 16 * var, reps=20)
            print("%s var=%d: %.4f ms (%.0f TF f16-issued equivalent)" % (tag, var, ms, 2 * fl / ms / 1e9), flush=True)
