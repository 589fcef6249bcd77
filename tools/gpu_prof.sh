#!/bin/bash
# kernel trace of a 450-restart step at the default (wide) batch width -> gpurun_out/${PROF_OUT:-r3_kernel_stats.txt}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof
( cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o trace -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --restarts-per-k ${RPK:-50} > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1 )
DB=$(ls gpurun_out/prof/*/*results.db gpurun_out/prof/*results.db 2>/dev/null | head -1)
python tools/export_profile.py $DB gpurun_out/${PROF_OUT:-r3_kernel_stats.txt} "python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --restarts-per-k ${RPK:-50} (${PROF_TAG:-default path, auto width}), $((9*${RPK:-50})) restarts" | head -24 | cut -c1-70,111-160
python tools/gap_analysis.py $DB gemm2h | head -8
