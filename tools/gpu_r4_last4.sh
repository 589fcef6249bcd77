#!/bin/bash
# the rebuilt final library once more on the GPU: one sparse-path test and the default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 150 python bench.py --no-cpu-baseline > gpurun_out/r4_bench_last4.json 2> gpurun_out/r4_bench_last4.err; echo "bench rc=$?"
python - <<P
import json
d = json.loads(open("gpurun_out/r4_bench_last4.json").read().strip().splitlines()[-1])
print("bench:", round(d["value"], 1), "restarts/s; roofline", round(d["roofline"]["frac"], 3))
P
