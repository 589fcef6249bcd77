#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/profc; mkdir -p $R/gpurun_out/profc
cd /tmp && CPU=0 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $R/gpurun_out/profc -o trace -- python $R/tools/gpu_cons.py > $R/gpurun_out/profc.log 2>&1
tail -5 $R/gpurun_out/profc.log
