#!/bin/bash
for m in 3; do echo "== probe mode $m"; CNMF_GEMM3=$m timeout 300 python tools/probe_gemm3.py 2>&1 | grep -v amdgpu.ids | grep -v "f32 MFMA kernel"; done
CNMF_GEMM3=3 timeout 900 python -m pytest tests/test_gpu_nmf.py tests/test_gpu_configs.py tests/test_gpu_edges.py -m gpu -x -q 2>&1 | grep "passed\|failed\|Error" | tail -3
for m in 2 3; do
CNMF_GEMM3=$m timeout 600 python bench.py --no-cpu-baseline --restarts-per-k 10 --steps 1 --warmup 1 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('GEMM3=$m restarts/s %.2f' % d['value'], 'riter/s %.0f' % d['config']['restart_iterations_per_s'], 'passA %.3f ms passB %.3f ms' % (r['avg_launch_ms']['passA'], r['avg_launch_ms']['passB']), 'gemm share %.3f' % r['gemm_share_of_gpu_time'])"
done
