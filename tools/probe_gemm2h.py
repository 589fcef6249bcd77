"""Probe of the f16 two-plane count GEMM: accuracy vs float64 and throughput next to the three-plane bf16 kernel;
instruction-stream variants (nsub | var << 4)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cnmf_amd.engine import Engine
eng = Engine(0)
rs = np.random.RandomState(0)
for K, J, ns in [(64, 40, 1), (2048, 1000, 1), (4096, 520, 4)]:
    A = np.abs(rs.standard_normal((256, K)) * np.exp(rs.standard_normal((256, K)))).astype(np.float32)
    B = rs.poisson(3.0, size=(J, K)).astype(np.float32); B[0, :3] = [2048, 255, 0]
    ref = A.astype(np.float64) @ B.astype(np.float64).T
    base = None
    for nsub in (1, 2, 2 + 16, 2 + 64, 2 + 80):
        C, _ = eng.debug_gemm2h(A, B, nsplit=ns, nsub=nsub)
        if base is None: base = C
        print("acc2h K=%d J=%d ns=%d nsub=%d var=%d: maxerr/max %.3e  rel-to-sum %.3e  same-as-nsub1 %s" % (K, J, ns, nsub & 15, nsub >> 4, np.abs(C - ref).max() / np.abs(ref).max(), (np.abs(C - ref) / np.maximum(ref, 1e-30)).max(), np.array_equal(C, base)), flush=True)
for K, J, ns, tag in [(2048, 50176, 1, "passA-shape (196 tiles)"), (50176, 2048, 32, "passB ns32")]:
    A = rs.rand(256, K).astype(np.float32)
    B = (rs.poisson(1.0, size=(J, K))).astype(np.float32)
    fl = 2.0 * 256 * K * J
    for rep in range(2):
        for nsub in (2 + 16, 2 + 64, 2 + 80):
            _, ms = eng.debug_gemm2h(A, B, nsplit=ns, nsub=nsub, reps=20)
            print("%s gemm2h nsub=%d var=%d: %.4f ms -> %.1f TF f32-equivalent (%.0f TF f16 issued)" % (tag, nsub & 15, nsub >> 4, ms, fl / ms / 1e9, 2 * fl / ms / 1e9), flush=True)
