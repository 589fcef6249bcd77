#!/bin/bash
# per-tile partials (step 1 of the fused W half-step): parity + determinism + bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4_pytest.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r4_pytest.log
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r4_bench_step1.json 2> gpurun_out/r4_bench_step1.err
python - <<P
import json
d = json.loads(open("gpurun_out/r4_bench_step1.json").read().strip().splitlines()[-1])
print("bench:", round(d["value"], 1), "restarts/s; roofline", round(d["roofline"]["frac"], 3), "e2e", round(d["roofline"]["end_to_end"]["frac"], 3), "gemm share", round(d["roofline"]["gemm_share_of_gpu_time"], 3), "tail", round(d["config"]["tail"]["share_of_gpu_time"], 3))
P
