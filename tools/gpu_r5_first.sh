#!/bin/bash
# Round-5 first GPU session: the new Kullback-Leibler tail tests + the suites its plumbing touches, then the kernel trace of a
# 450-restart step (W sweep back on the round-3 partial layout: target <= 156 us per 1024-column launch).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kl_tail.py -x -q -s > gpurun_out/r5_kl_tail.log 2>&1; echo "kl_tail rc=$?" | tee gpurun_out/r5_first.status
grep -v "^$" gpurun_out/r5_kl_tail.log | tail -25
timeout 1200 python -m pytest tests/test_gpu_mu_sparse.py tests/test_gpu_mu.py tests/test_gpu_pipeline.py tests/test_gpu_tail.py tests/test_gpu_determinism.py tests/test_gpu_nmf.py tests/test_gpu_edges.py -x -q > gpurun_out/r5_suite_a.log 2>&1; echo "suite_a rc=$?" | tee -a gpurun_out/r5_first.status
tail -5 gpurun_out/r5_suite_a.log
PROF_TAG="default path, auto width" RPK=50 PROF_OUT=r5_kernel_stats_first.txt bash tools/gpu_prof.sh > gpurun_out/r5_prof.log 2>&1; rm -rf gpurun_out/prof
head -14 gpurun_out/r5_kernel_stats_first.txt | cut -c1-90,111-170
tail -3 gpurun_out/prof_bench.log | cut -c1-400
