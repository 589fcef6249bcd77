#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
for arm in store nostore; do
  if [ $arm = nostore ]; then export CNMF_G2_NOSTORE=1; else unset CNMF_G2_NOSTORE; fi
  rm -rf gpurun_out/prof_s
  ( cd /tmp && AB_ITERS=30 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_s -o trace -- python $GRAFT_REPO_ROOT/tools/fused_ab.py child x$arm > $GRAFT_REPO_ROOT/gpurun_out/prof_s.log 2>&1 )
  DB=$(ls gpurun_out/prof_s/*/*results.db gpurun_out/prof_s/*results.db 2>/dev/null | head -1)
  echo "== $arm"
  python tools/export_profile.py $DB gpurun_out/r4_store_$arm.txt "fused_ab child ($arm)" | head -8 | cut -c1-60,100-170
  rm -rf gpurun_out/prof_s gpurun_out/fused_ab_x$arm.npz
done
