#!/bin/bash
# round 4, third GPU session: the whole GPU suite on the new tree, then the default bench with and without the learned
# queue prior (CNMF_NO_PRIOR=1)
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/r4_gpu_tests.log
tail -15 gpurun_out/r4_gpu_tests.log
for v in prior noprior; do
  if [ $v = noprior ]; then export CNMF_NO_PRIOR=1; else unset CNMF_NO_PRIOR; fi
  timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r4_bench_$v.json 2> gpurun_out/r4_bench_$v.err
  python - <<P
import json
try:
    d = json.loads(open("gpurun_out/r4_bench_$v.json").read().strip().splitlines()[-1])
    print("$v:", round(d["value"], 1), "restarts/s", round(d["ms_per_step"]), "ms", "tail", d["config"].get("tail"), "util", d["config"].get("column_utilisation"))
except Exception as e:
    print("$v failed", e); print(open("gpurun_out/r4_bench_$v.err").read()[-1500:])
P
done
