#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fused.py -m gpu -x -q 2>&1 | tail -4
timeout 600 python tools/profile_e2e_host.py > gpurun_out/r4_e2e_hostprof.txt 2>&1
grep -n "====" -A26 gpurun_out/r4_e2e_hostprof.txt | grep -v "^--" | cut -c1-150 | head -120
