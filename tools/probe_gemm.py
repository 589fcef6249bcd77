"""GPU probe: time the MFMA GEMM variants at the north-star shapes (not a test)."""
import json
import os
import sys


import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cnmf_amd.engine import Engine  # noqa: E402


def main():
    N = int(os.environ.get("PROBE_N", 50048))
    G = 2016
    eng = Engine(0)
    rs = np.random.RandomState(0)
    X = rs.standard_normal((N, G)).astype(np.float32)
    res = []
    for KC in (32, 64, 128, 256):
        H = rs.standard_normal((KC, G)).astype(np.float32)
        Wt = rs.standard_normal((KC, N)).astype(np.float32)
        for variant in (1, 2, 3):
            _, ms = eng.debug_gemm(0, H, X, variant=variant, reps=5)
            fl = 2.0 * N * G * KC
            res.append(dict(mode="A", KC=KC, variant=variant, ms=ms, tflops=fl / ms / 1e9,
                            x_gbs=N * G * 4 / ms / 1e6))
            print(res[-1], flush=True)
        for variant in (1, 2, 3):
            for ns in (8, 16, 32, 64):
                _, ms = eng.debug_gemm(1, Wt, X, variant=variant, nsplit=ns, reps=5)
                fl = 2.0 * N * G * KC
                res.append(dict(mode="B", KC=KC, variant=variant, nsplit=ns, ms=ms,
                                tflops=fl / ms / 1e9, x_gbs=N * G * 4 / ms / 1e6))
                print(res[-1], flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/probe_gemm.json", "w"), indent=1)


if __name__ == "__main__":
    main()
