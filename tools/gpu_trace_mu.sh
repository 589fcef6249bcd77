#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/profm; mkdir -p $R/gpurun_out/profm
cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/profm -o trace -- python $R/tools/gpu_mu_probe.py > $R/gpurun_out/profm.log 2>&1
tail -4 $R/gpurun_out/profm.log
