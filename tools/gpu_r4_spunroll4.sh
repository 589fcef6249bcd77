#!/bin/bash
# four gathers in flight at ranks 13..16 as well: parity test at rank 16 / 14, same-box A/B of two builds at ranks 13 and 16
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 60 python -m pytest tests/test_gpu_mu_sparse.py -x -q -k "1200-900-16" 2>&1 | tail -2
for lib in prev new; do
  echo "build $lib"
  if [ $lib = prev ]; then export CNMF_LIB_PATH=$GRAFT_REPO_ROOT/cnmf_amd/libcnmf_hip_prev.so; else unset CNMF_LIB_PATH; fi
  SP_CELLS=100000 SP_ONLY=1 SP_KS=13x32,16x32,13x32,16x32 SP_MODES=1 MU_ITERS=100 timeout 100 python tools/mu_sparse_probe.py 2>&1 | grep "us per"
done > gpurun_out/r4_mu_sparse_unroll4_ab.txt 2>&1
cat gpurun_out/r4_mu_sparse_unroll4_ab.txt
