#!/bin/bash
# Round-2 profile session: kernel trace of one short bench step (180 restarts) -> per-kernel stats + gap analysis
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof
( cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o trace -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --restarts-per-k 20 > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1 )
DB=$(ls gpurun_out/prof/*/*results.db gpurun_out/prof/*results.db 2>/dev/null | head -1)
echo "db: $DB"
python tools/export_profile.py $DB gpurun_out/kernel_stats.txt "python bench.py --steps 1 --warmup 0 --no-cpu-baseline --restarts-per-k 20 (${PROF_TAG:-default path}), 180 restarts" | head -30
python tools/gap_analysis.py $DB gemm2h | head -12
