#!/bin/bash
python -m pytest tests/test_gpu_nmf.py tests/test_gpu_configs.py tests/test_gpu_edges.py tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | grep "passed\|failed\|rror" | tail -3
for e in "" "CNMF_NO_DEFRAG=1"; do
env $e CNMF_DEBUG=1 python bench.py --no-cpu-baseline --restarts-per-k 30 --steps 1 --warmup 0 2>gpurun_out/err_d.log | python -c "
import sys, json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$e restarts/s %.2f' % d['value'], 'riter/s %.0f' % d['config']['restart_iterations_per_s'], 'util %.4f' % d['config']['column_utilisation'], 'mean_it %.1f' % d['config']['mean_iterations_per_restart'])"
grep "defrag\|KC=256" gpurun_out/err_d.log | tail -2
done
