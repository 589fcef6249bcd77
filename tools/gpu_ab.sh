#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; grep -n "passed\|failed\|Error" gpurun_out/gpu_tests.log | tail -5
python bench.py --no-cpu-baseline --restarts-per-k 10 --steps 1 --warmup 1 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('restarts/s %.2f' % d['value'], 'riter/s %.0f' % d['config']['restart_iterations_per_s'], 'passA %.3f ms passB %.3f ms' % (r['avg_launch_ms']['passA'], r['avg_launch_ms']['passB']), 'gemm share %.3f' % r['gemm_share_of_gpu_time'])"
