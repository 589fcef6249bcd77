#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4_pytest.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r4_pytest.log
for i in 1 2; do
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r4_bench_x.json 2> gpurun_out/r4_bench_x.err
python - <<P
import json
d = json.loads(open("gpurun_out/r4_bench_x.json").read().strip().splitlines()[-1])
print("bench:", round(d["value"], 1), "restarts/s; roofline", round(d["roofline"]["frac"], 3), "passA/B TF", round(d["roofline"]["achieved_passA"]), round(d["roofline"]["achieved_passB"]), "e2e", round(d["roofline"]["end_to_end"]["frac"], 3), "gemm share", round(d["roofline"]["gemm_share_of_gpu_time"], 3), "tail", round(d["config"]["tail"]["share_of_gpu_time"], 3))
P
done
