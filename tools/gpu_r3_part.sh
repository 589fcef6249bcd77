#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_nmf.py tests/test_gpu_golden_big.py tests/test_gpu_configs.py tests/test_gpu_determinism.py -m gpu -x -q > gpurun_out/r3_part_pytest.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r3_part_pytest.log
for np in 0 1; do
  if [ $np = 1 ]; then export CNMF_NO_PART=1; else unset CNMF_NO_PART; fi
  for em in "0/1" "0/8"; do
    timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --emulate-rank $em 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; print('no_part=$np shard $em: %.1f restarts/s util %.3f tail %.0f ms (%d its, mean live %.0f)' % (d['value'], c['column_utilisation'], c['tail']['ms_per_step'], c['tail']['iterations_per_step'], c['tail']['mean_live_columns']))"
  done
done
