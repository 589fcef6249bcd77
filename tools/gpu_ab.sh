#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/profg; mkdir -p $R/gpurun_out/profg
( cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/profg -o trace -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --restarts-per-k 10 > $R/gpurun_out/profg_bench.log 2>&1 )
python $R/tools/gap_analysis.py $R/gpurun_out/profg/trace_results.db gemm3c | head -3
python $R/tools/export_profile.py $R/gpurun_out/profg/trace_results.db /tmp/x.txt t | cut -c1-60,110-160 | head -12
rm -rf $R/gpurun_out/profg
