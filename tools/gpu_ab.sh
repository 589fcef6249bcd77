#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python bench.py --steps 1 --warmup 0 --restarts-per-k 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('bench', 'restarts/s %.2f' % d['value'], 'riter/s %.0f' % d['config']['restart_iterations_per_s'], 'passA %.3f ms passB %.3f ms' % (r['avg_launch_ms']['passA'], r['avg_launch_ms']['passB']), 'TF A %.1f B %.1f' % (r['achieved_passA'], r['achieved_passB']), 'gemm share %.3f util %.3f' % (r['gemm_share_of_gpu_time'], d['config']['column_utilisation']))
"
