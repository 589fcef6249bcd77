#!/bin/bash
# kernel trace of the KL multiplicative-update probe (matrix-pipe path)
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/profm2
MU_ITERS=${MU_ITERS:-30} timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/profm2 -o trace -- python tools/gpu_mu_probe.py > gpurun_out/profm2.log 2>&1
grep "KL k" gpurun_out/profm2.log
python tools/export_profile.py $(find gpurun_out/profm2 -name "*.db" | head -1) gpurun_out/mu_stats.txt "tools/gpu_mu_probe.py (KL, C3, matrix-pipe path)" | head -24
