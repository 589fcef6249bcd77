#!/bin/bash
# Full GPU-box session for the round's final numbers: tests, smoke, default bench, e2e, kernel-trace.
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; grep -n "passed\|failed\|rror" gpurun_out/pytest_gpu.log | tail -4
python __graft_entry__.py smoke 2>&1 | tail -2 | tee gpurun_out/smoke.log
python bench.py 2> gpurun_out/bench.err > gpurun_out/bench.json; cut -c1-300 gpurun_out/bench.json
python tools/e2e_c3.py > gpurun_out/e2e_c3.json 2> gpurun_out/e2e.err; tail -c 600 gpurun_out/e2e_c3.json
( cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o trace -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1 )
