#!/bin/bash
# GPU-box session for the round's final numbers: smoke, default bench, e2e, kernel-trace (+ optionally the tests).
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ -n "$DO_TESTS" ]; then python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; grep -n "passed\|failed\|rror" gpurun_out/pytest_gpu.log | tail -4; fi
python __graft_entry__.py smoke 2>&1 | tail -2 | tee gpurun_out/smoke.log
python bench.py 2> gpurun_out/bench.err > gpurun_out/bench.json; cut -c1-300 gpurun_out/bench.json
rm -f gpurun_out/e2e_c3.json; python tools/e2e_c3.py > gpurun_out/e2e_c3.json 2> gpurun_out/e2e.err; head -c 700 gpurun_out/e2e_c3.json; echo
rm -rf gpurun_out/prof
( cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o trace -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --restarts-per-k 20 > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1 )
python tools/gap_analysis.py gpurun_out/prof/trace_results.db gemm3c | head -4
