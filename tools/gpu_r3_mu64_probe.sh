#!/bin/bash
# time per restart-iteration of the KL solver at C3 for ranks 32 / 40 / 64 (matrix pipe) and 40 / 64 on the vector-ALU kernels
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python - > gpurun_out/r3_mu64_probe.txt 2>&1 <<'PY'
import os, sys, time
import numpy as np
sys.path.insert(0, '.')
from cnmf_amd import synth
from cnmf_amd.engine import Engine
X = synth.make_config("C3", dtype=np.float32)
eng = Engine(0); eng.set_matrix(X)
eng.nmf_mu_batch([5], seeds=[1], max_iter=3, tol=0, warn=False)
def run(ks, its, tag):
    out = []
    for it in (its, 2 * its):                    # two lengths: the difference is free of the per-restart set-up cost
        t = time.perf_counter()
        H, _, n, err = eng.nmf_mu_batch(ks, seeds=list(range(7, 7 + len(ks))), max_iter=it, tol=0, warn=False)
        out.append(time.perf_counter() - t)
    per = (out[1] - out[0]) / (its * len(ks)) * 1e6
    print("%s KL k=%s x%d: %d / %d iterations in %.3f / %.3f s -> %.1f us per restart-iteration (set-up excluded)"
          % (tag, sorted(set(ks)), len(ks), its, 2 * its, out[0], out[1], per), flush=True)
for ks in ([9] * 16, [20] * 16, [32] * 16, [40] * 16, [64] * 16, [64] * 32):
    run(ks, 50, "matrix pipe")
os.environ["CNMF_MU_VALU"] = "1"
for ks in ([40] * 1, [64] * 1):
    run(ks, 20, "vector ALU ")
PY
cat gpurun_out/r3_mu64_probe.txt
