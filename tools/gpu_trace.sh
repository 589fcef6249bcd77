#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof; mkdir -p $R/gpurun_out/prof
cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o trace -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras ${BENCH_ARGS} > $R/gpurun_out/prof_bench.log 2>&1
tail -2 $R/gpurun_out/prof_bench.log | cut -c1-600
