#!/bin/bash
# MU ranks 33..64 on the matrix pipe: parity tests, then time per restart-iteration at C3 (matrix pipe vs the vector-ALU kernels)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_mu.py tests/test_gpu_edges.py -m gpu -x -q > gpurun_out/r3_mu64_pytest.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/r3_mu64_pytest.log
timeout 600 python - > gpurun_out/r3_mu64_probe.txt 2>&1 <<'PY'
import os, sys, time
import numpy as np
sys.path.insert(0, '.')
from cnmf_amd import synth
from cnmf_amd.engine import Engine
X = synth.make_config("C3", dtype=np.float32)
eng = Engine(0); eng.set_matrix(X)
eng.nmf_mu_batch([5], seeds=[1], max_iter=3, tol=0, warn=False)
def run(ks, its, tag):
    t = time.perf_counter()
    H, _, n, err = eng.nmf_mu_batch(ks, seeds=list(range(7, 7 + len(ks))), max_iter=its, tol=0, warn=False)
    dt = time.perf_counter() - t
    print("%s KL k=%s x%d: %d iterations each in %.3f s -> %.1f us per restart-iteration" % (tag, sorted(set(ks)), len(ks), n[0], dt, dt / n.sum() * 1e6), flush=True)
for ks in ([20] * 16, [32] * 16, [40] * 1, [40] * 8, [40] * 16, [64] * 16):
    run(ks, 20, "matrix pipe")
os.environ["CNMF_MU_VALU"] = "1"
for ks in ([40] * 2, [64] * 2):
    run(ks, 10, "vector ALU ")
PY
cat gpurun_out/r3_mu64_probe.txt
