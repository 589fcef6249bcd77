#!/bin/bash
# same-box A/B of the queue forecast (restart length predicted from the violation decay) vs ages only
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1; do
for arm in forecast ages; do
  if [ $arm = ages ]; then export CNMF_NO_FORECAST=1; else unset CNMF_NO_FORECAST; fi
  timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r4_bench_x.json 2> gpurun_out/r4_bench_x.err
  python - <<P
import json
d = json.loads(open("gpurun_out/r4_bench_x.json").read().strip().splitlines()[-1])
print("$arm:", round(d["value"], 1), "restarts/s; tail", round(d["config"]["tail"]["share_of_gpu_time"], 3), "util", round(d["config"]["column_utilisation"], 4), "e2e", round(d["roofline"]["end_to_end"]["frac"], 3))
P
done
done
unset CNMF_NO_FORECAST
timeout 900 python -m pytest tests/test_gpu_determinism.py tests/test_gpu_nmf.py tests/test_gpu_golden_big.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -2
