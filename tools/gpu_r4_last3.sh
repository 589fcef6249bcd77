#!/bin/bash
# the two tests of the non-zero path the unroll A/B session left out, and smoke, on the final build
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_gpu_mu_sparse.py -x -q -k "2500-1300-24 or selection" > gpurun_out/r4_last3_tests.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r4_last3_tests.log
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
