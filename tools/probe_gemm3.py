"""Probe of the split-operand bf16 MFMA GEMM: accuracy vs float64 and throughput vs the f32 MFMA kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cnmf_amd.engine import Engine

eng = Engine(0)
rs = np.random.RandomState(0)
# accuracy
for KC, K, J, ns in [(256, 64, 192, 1), (256, 2016, 992, 1), (256, 4096, 320, 4)]:
    A = (rs.standard_normal((KC, K)) * np.exp(rs.standard_normal((KC, K)))).astype(np.float32)
    B = (rs.standard_normal((J, K)) * np.exp(rs.standard_normal((J, K)))).astype(np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64).T
    C3, _ = eng.debug_gemm3(A, B, nsplit=ns)
    C1, _ = eng.debug_gemm(0, A, B, variant=2)
    s = np.abs(ref).max()
    print("acc KC=%d K=%d J=%d ns=%d: gemm3 maxerr/max %.3e  f32 MFMA %.3e   | rel-to-|a||b| gemm3 %.3e f32 %.3e" % (
        KC, K, J, ns, np.abs(C3 - ref).max() / s, np.abs(C1 - ref).max() / s,
        (np.abs(C3 - ref) / (np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64).T)).max(),
        (np.abs(C1 - ref) / (np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64).T)).max()))
# non-negative operands like the engine's (X >= 0, H >= 0)
A = np.abs(rs.standard_normal((256, 2016))).astype(np.float32)
B = (np.abs(rs.standard_normal((1984, 2016))) * (rs.rand(1984, 2016) < 0.3)).astype(np.float32)
ref = A.astype(np.float64) @ B.astype(np.float64).T
C3, _ = eng.debug_gemm3(A, B); C1, _ = eng.debug_gemm(0, A, B, variant=2)
print("nonneg: gemm3 max rel err %.3e   f32 MFMA %.3e" % ((np.abs(C3 - ref) / ref.clip(1e-30)).max(), (np.abs(C1 - ref) / ref.clip(1e-30)).max()))
# throughput
for KC, K, J, ns, tag in [(256, 2016, 50048, 1, "passA"), (256, 50048, 2048, 32, "passB ns32"), (256, 50048, 2048, 16, "passB ns16")]:
    A = rs.rand(KC, K).astype(np.float32)
    B = rs.rand(J, K).astype(np.float32)
    _, ms = eng.debug_gemm3(A, B, nsplit=ns, reps=10)
    fl = 2.0 * KC * K * J
    print("%s gemm3 %dx%dx%d: %.3f ms  -> %.1f TF f32-equivalent (%.0f TF bf16 issued)" % (tag, KC, K, J, ms, fl / ms / 1e9, 6 * fl / ms / 1e9))
    if tag == "passA":
        _, ms1 = eng.debug_gemm(0, A, B, variant=2, reps=10)
        print("   f32 MFMA kernel: %.3f ms -> %.1f TF" % (ms1, fl / ms1 / 1e9))
