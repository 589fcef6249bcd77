#!/bin/bash
for m in 1 2; do echo "== probe mode $m"; CNMF_GEMM3=$m python tools/probe_gemm3.py 2>&1 | grep -v amdgpu.ids | grep -v "f32 MFMA kernel"; done
python -m pytest tests/test_gpu_nmf.py -m gpu -x -q 2>&1 | grep "passed\|failed" | tail -3
for m in 2 2; do
CNMF_GEMM3=$m python bench.py --no-cpu-baseline --restarts-per-k 10 --steps 1 --warmup 1 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('GEMM3=$m restarts/s %.2f' % d['value'], 'riter/s %.0f' % d['config']['restart_iterations_per_s'], 'passA %.3f ms passB %.3f ms' % (r['avg_launch_ms']['passA'], r['avg_launch_ms']['passB']), 'gemm share %.3f' % r['gemm_share_of_gpu_time'])"
done
