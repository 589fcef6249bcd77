#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
for arm in 1 0; do
  rm -rf gpurun_out/prof_f$arm
  ( cd /tmp && CNMF_FUSE_A=$arm rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_f$arm -o trace -- python $GRAFT_REPO_ROOT/tools/fused_ab.py child arm$arm > $GRAFT_REPO_ROOT/gpurun_out/prof_f$arm.log 2>&1 )
  DB=$(ls gpurun_out/prof_f$arm/*/*results.db gpurun_out/prof_f$arm/*results.db 2>/dev/null | head -1)
  echo "== CNMF_FUSE_A=$arm"
  python tools/export_profile.py $DB gpurun_out/r4_fused_arm$arm.txt "fused_ab child, CNMF_FUSE_A=$arm" | head -12 | cut -c1-60,100-170
  rm -rf gpurun_out/prof_f$arm gpurun_out/fused_ab_arm$arm.npz
done
