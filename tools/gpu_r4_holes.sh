#!/bin/bash
# same-box A/B of the refill rule: holes only for short restarts + re-packing for the head of the queue (default) vs filling
# every hole at once with the largest pending rank that fits (CNMF_HOLES=fill, rounds 1-3)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
for arm in wait fill; do
  if [ $arm = fill ]; then export CNMF_HOLES=fill; else unset CNMF_HOLES; fi
  CNMF_DEBUG=1 timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r4_bench_x.json 2> gpurun_out/r4_bench_x.err
  python - <<P
import json
d = json.loads(open("gpurun_out/r4_bench_x.json").read().strip().splitlines()[-1])
print("$arm:", round(d["value"], 1), "restarts/s; tail", round(d["config"]["tail"]["share_of_gpu_time"], 3), "util", round(d["config"]["column_utilisation"], 4), "e2e", round(d["roofline"]["end_to_end"]["frac"], 3))
P
  grep "defragmentations" gpurun_out/r4_bench_x.err | tail -1
done
done
unset CNMF_HOLES
timeout 900 python -m pytest tests/test_gpu_determinism.py tests/test_gpu_nmf.py tests/test_gpu_golden_big.py tests/test_gpu_configs.py tests/test_gpu_edges.py -m gpu -x -q 2>&1 | tail -2
