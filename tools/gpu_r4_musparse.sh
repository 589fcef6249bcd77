#!/bin/bash
# Kullback-Leibler on the non-zeros: parity tests, the timing probe at config 4's shape, a kernel trace of its sparse leg
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_mu_sparse.py tests/test_gpu_mu.py -x -q > gpurun_out/r4_musparse_tests.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r4_musparse_tests.log
CNMF_DEBUG=1 timeout 900 python tools/mu_sparse_probe.py > gpurun_out/r4_mu_sparse_probe.txt 2>&1; echo "probe rc=$?"
grep -v "^\[cnmf\] batch" gpurun_out/r4_mu_sparse_probe.txt | grep "non-zero\|us per" 
bash tools/gpu_r4_musparse_trace.sh
