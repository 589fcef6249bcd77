#!/bin/bash
mkdir -p gpurun_out
for cfg in "base:" "bk16x768:CNMF_BK16=1,CNMF_SK_WGS=768" "bk16x1024:CNMF_BK16=1,CNMF_SK_WGS=1024" "bk32x768:CNMF_SK_WGS=768"; do
  tag=${cfg%%:*}; envs=$(echo ${cfg#*:} | tr ',' ' ')
  env $envs python bench.py --steps 1 --warmup 0 --restarts-per-k 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$tag', 'restarts/s %.2f' % d['value'], 'riter/s %.0f' % d['config']['restart_iterations_per_s'], 'passA %.3f ms passB %.3f ms' % (r['avg_launch_ms']['passA'], r['avg_launch_ms']['passB']), 'TF A %.1f B %.1f' % (r['achieved_passA'], r['achieved_passB']), 'gemm share %.3f util %.3f' % (r['gemm_share_of_gpu_time'], d['config']['column_utilisation']))
"
done
