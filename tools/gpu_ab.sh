#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_nmf.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | grep "passed\|failed\|rror\|assert\|^E" | tail -12
timeout 600 python bench.py --no-cpu-baseline --restarts-per-k 10 --steps 1 --warmup 1 2>gpurun_out/err_c.log | python -c "
import sys, json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('restarts/s %.2f' % d['value'], 'passA %.3f ms passB %.3f ms' % (r['avg_launch_ms']['passA'], r['avg_launch_ms']['passB']))" || tail -5 gpurun_out/err_c.log
