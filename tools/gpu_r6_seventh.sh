#!/bin/bash
# Round-6 seventh GPU session: whole-bench A/B of process-lifetime knobs on one box (the launches of an iteration share one
# power budget: a pass-level result need not carry over): the XCD mapping of pass A on the COUNT path.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
  for xm in 1 0; do
    CNMF_G2_XMAP=$xm timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r6_xm.json 2>> gpurun_out/r6_xm.err
    python - <<P
import json
d = json.loads(open("gpurun_out/r6_xm.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("count path CNMF_G2_XMAP=$xm rep $rep:", round(d["value"], 1), "restarts/s; e2e", round(r["end_to_end"]["frac"], 4), "pass A/B ms", {k: round(v, 4) for k, v in r["avg_launch_ms"].items()})
P
  done
done 2>&1 | tee gpurun_out/r6_count_xmap_ab.txt
