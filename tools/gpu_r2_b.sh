#!/bin/bash
# Round-2 GPU session B: f16 two-plane GEMM probe, GPU tests (no -x), default bench.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/probe_gemm2h.py > gpurun_out/probe_gemm2h.log 2>&1; tail -30 gpurun_out/probe_gemm2h.log
timeout 900 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/pytest_gpu.log 2>&1
grep -n "passed\|failed\|Error\|error\|assert" gpurun_out/pytest_gpu.log | tail -40
timeout 600 python bench.py --steps 2 --warmup 1 2> gpurun_out/bench.err > gpurun_out/bench.json; cut -c1-1500 gpurun_out/bench.json; tail -3 gpurun_out/bench.err
