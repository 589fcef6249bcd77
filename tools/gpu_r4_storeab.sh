#!/bin/bash
# same-box A/B: GEMM tiles stored through LDS with 16-byte stores (the tree's build) vs straight from the registers
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
for arm in lds direct; do
  if [ $arm = direct ]; then export CNMF_LIB_PATH=$GRAFT_REPO_ROOT/tools/bin/libcnmf_hip_directstore.so; else unset CNMF_LIB_PATH; fi
  timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r4_bench_x.json 2> gpurun_out/r4_bench_x.err
  python - <<P
import json
d = json.loads(open("gpurun_out/r4_bench_x.json").read().strip().splitlines()[-1])
print("$arm:", round(d["value"], 1), "restarts/s; passA/B TF", round(d["roofline"]["achieved_passA"]), round(d["roofline"]["achieved_passB"]), "e2e", round(d["roofline"]["end_to_end"]["frac"], 3))
P
done
done
