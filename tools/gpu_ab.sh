#!/bin/bash
for i in 1 2; do python -m pytest tests -m gpu -x -q 2>&1 | grep "passed\|failed\|rror" | tail -2; done
for cfg in "C2:--workload C2 --kmin 10 --kmax 10 --restarts-per-k 100" "C1:--workload C1 --kmin 7 --kmax 7 --restarts-per-k 20" "C3n10k:--workload C3 --n-cells 10000 --restarts-per-k 10"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  python bench.py --steps 1 --warmup 1 --no-cpu-baseline $args 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$tag', 'restarts/s %.2f' % d['value'], 'riter/s %.0f' % d['config']['restart_iterations_per_s'], 'passA %.3f ms passB %.3f ms' % (r['avg_launch_ms']['passA'], r['avg_launch_ms']['passB']), 'gemm share %.3f kc %d' % (r['gemm_share_of_gpu_time'], d['config']['packed_columns']), d['dtype'][:30])
"
done
