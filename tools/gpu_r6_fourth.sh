#!/bin/bash
# Round-6 fourth GPU session: Kullback-Leibler on the non-zeros with the entry stream two trips ahead (default build) against
# one trip ahead (tools/bin/libcnmf_pf1.so, -DCNMF_SP_PF=1), same box, alternating; 200 000 x 2 000 / 9 % and 50 000 x 2 000.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
  for lib in "" tools/bin/libcnmf_pf1.so; do
    echo "== lib=${lib:-default (PF=2)} rep $rep"
    CNMF_LIB_PATH=${lib:+$GRAFT_REPO_ROOT/$lib} SP_ONLY=1 SP_LONG=1 MU_ITERS=100 timeout 600 python tools/mu_sparse_probe.py 2>&1 | grep "us per"
    CNMF_LIB_PATH=${lib:+$GRAFT_REPO_ROOT/$lib} SP_CELLS=50000 SP_ONLY=1 SP_KS=9x36,5x36,13x36 MU_ITERS=100 timeout 600 python tools/mu_sparse_probe.py 2>&1 | grep "us per"
  done
done 2>&1 | tee gpurun_out/r6_mu_sparse_prefetch_ab.txt
