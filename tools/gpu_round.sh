#!/bin/bash
# One GPU-box session: tests, smoke, bench, kernel-trace profile.  Logs -> gpurun_out/
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.log
python __graft_entry__.py smoke 2>&1 | tail -3 | tee gpurun_out/smoke.log
python bench.py --steps ${BENCH_STEPS:-2} --warmup 1 2> gpurun_out/bench.err | tee gpurun_out/bench.json
tail -5 gpurun_out/bench.err
if [ -n "$DO_PROF" ]; then
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o trace -- \
      python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1 )
  ls -R gpurun_out/prof | head -20
fi
