#!/bin/bash
# four gathers in flight per lane (ranks up to 12) against two: parity tests on the new build, same-box A/B of two builds
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_mu_sparse.py -x -q -k "3000-700-7 or 1500-2300-12 or 130-70-3 or refit" > gpurun_out/r4_spunroll_tests.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r4_spunroll_tests.log
for lib in u2 u4 u2 u4; do
  echo "build $lib"
  if [ $lib = u2 ]; then export CNMF_LIB_PATH=$GRAFT_REPO_ROOT/cnmf_amd/libcnmf_hip_u2.so; else unset CNMF_LIB_PATH; fi
  SP_ONLY=1 SP_LONG=1 SP_MODES=1 MU_ITERS=150 timeout 600 python tools/mu_sparse_probe.py 2>&1 | grep "us per"
done > gpurun_out/r4_mu_sparse_unroll_ab.txt 2>&1
cat gpurun_out/r4_mu_sparse_unroll_ab.txt
