#!/bin/bash
# The driver's own command line for the N = 1 bench (BENCH_rNN.json): wall-clock around it and the line.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
t0=$(date +%s.%N)
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6_bench_steps20.json 2> gpurun_out/r6_bench_steps20.err; echo "bench rc=$?"
t1=$(date +%s.%N)
python - <<P
import json
d = json.loads(open("gpurun_out/r6_bench_steps20.json").read().strip().splitlines()[-1])
print("wall %.1f s;" % ($t1 - $t0), round(d["value"], 1), "restarts/s, steps", d["steps"], "warmup", d["warmup"], "ms/step", round(d["ms_per_step"], 1), "-> timed region %.1f s;" % (d["steps"] * d["ms_per_step"] / 1e3), "frac", round(d["roofline"]["frac"], 3), "e2e", round(d["roofline"]["end_to_end"]["frac"], 3), d["config"]["regime"], "general", round(d["general_path"]["restarts_per_s"], 1), "e2e", round(d["e2e"]["total_s"], 2))
P
