#!/bin/bash
for e in "" "CNMF_F32_TAIL=1" "" "CNMF_F32_TAIL=1"; do
env $e python bench.py --no-cpu-baseline --restarts-per-k 20 --steps 1 --warmup 1 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$e restarts/s %.2f' % d['value'], 'riter/s %.0f' % d['config']['restart_iterations_per_s'], 'util %.4f' % d['config']['column_utilisation'])"
done
