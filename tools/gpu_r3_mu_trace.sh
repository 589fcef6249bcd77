#!/bin/bash
# kernel trace of the KL solver at padded rank 64 (16 restarts of rank 40, 16 of rank 64, 50 iterations each, C3)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
cat > /tmp/mu64_trace.py <<'PY'
import os, sys
import numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from cnmf_amd import synth
from cnmf_amd.engine import Engine
X = synth.make_config("C3", dtype=np.float32)
eng = Engine(0); eng.set_matrix(X)
eng.nmf_mu_batch([5], seeds=[1], max_iter=3, tol=0, warn=False)
for ks in ([40] * 16, [64] * 16):
    eng.nmf_mu_batch(ks, seeds=list(range(7, 7 + len(ks))), max_iter=50, tol=0, warn=False)
PY
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_mu -o mu -- python /tmp/mu64_trace.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
DB=$(ls gpurun_out/prof_mu/*/*results.db gpurun_out/prof_mu/*results.db 2>/dev/null | head -1)
python tools/export_profile.py $DB gpurun_out/r3_kernel_stats_mu_k64.txt "tools/gpu_r3_mu_trace.sh: KL multiplicative updates at C3 (50000 x 2000), 16 restarts of rank 40 and 16 of rank 64, 50 iterations each (padded rank 64: two restarts per workgroup, two M tiles)" > /dev/null 2>&1
rm -rf gpurun_out/prof_mu
head -16 gpurun_out/r3_kernel_stats_mu_k64.txt | cut -c1-100,112-160
