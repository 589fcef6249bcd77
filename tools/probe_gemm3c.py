"""Probe of the count-path GEMM (one integer plane for B): accuracy vs float64 and throughput."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cnmf_amd.engine import Engine
eng = Engine(0)
rs = np.random.RandomState(0)
for K, J, ns in [(16, 40, 1), (64, 300, 1), (2048, 1000, 1), (4096, 520, 4), (2048, 130, 7)]:
    A = (rs.standard_normal((256, K)) * np.exp(rs.standard_normal((256, K)))).astype(np.float32)
    B = rs.poisson(3.0, size=(J, K)).astype(np.float32); B[0, :3] = [256, 255, 0]
    ref = A.astype(np.float64) @ B.astype(np.float64).T
    C, _ = eng.debug_gemm3c(A, B, nsplit=ns)
    scale = np.abs(A).astype(np.float64) @ B.astype(np.float64).T
    print("acc K=%d J=%d ns=%d: maxerr/max %.3e  rel-to-|a||b| %.3e" % (K, J, ns, np.abs(C - ref).max() / np.abs(ref).max(), (np.abs(C - ref) / np.maximum(scale, 1e-30)).max()))
for K, J, ns, tag in [(2048, 50176, 1, "passA"), (50176, 2048, 32, "passB ns32")]:
    A = rs.rand(256, K).astype(np.float32)
    B = (rs.poisson(1.0, size=(J, K))).astype(np.float32)
    _, ms = eng.debug_gemm3c(A, B, nsplit=ns, reps=10)
    fl = 2.0 * 256 * K * J
    print("%s gemm3c 256x%dx%d: %.3f ms -> %.1f TF f32-equivalent (%.0f TF bf16 issued)" % (tag, K, J, ms, fl / ms / 1e9, 3 * fl / ms / 1e9))
