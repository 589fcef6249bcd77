#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_nmf.py tests/test_gpu_configs.py tests/test_gpu_edges.py -m gpu -x -q 2>&1 | tail -3
for cfg in "C2:--workload C2 --kmin 10 --kmax 10 --restarts-per-k 100" "C3n10k:--workload C3 --n-cells 10000 --restarts-per-k 10" "C3n25k:--workload C3 --n-cells 25000 --restarts-per-k 10"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  python bench.py --steps 1 --warmup 1 --no-cpu-baseline $args 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$tag', d['metric'], 'restarts/s %.2f' % d['value'], 'riter/s %.0f' % d['config']['restart_iterations_per_s'], 'mean_it %.0f' % d['config']['mean_iterations_per_restart'], 'passA %.3f ms passB %.3f ms' % (r['avg_launch_ms']['passA'], r['avg_launch_ms']['passB']), 'TF A %.1f B %.1f' % (r['achieved_passA'], r['achieved_passB']), 'gemm share %.3f util %.3f kc %d' % (r['gemm_share_of_gpu_time'], d['config']['column_utilisation'], d['config']['packed_columns']))
"
done
