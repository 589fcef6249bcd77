#!/bin/bash
# Round-6 third GPU session: the count path on the one-block spread stream (CNMF_G2_NSUB=1) against the default two-block one.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
for ns in 2 1 2 1; do
  CNMF_G2_NSUB=$ns timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r6_count_nsub${ns}.json 2>> gpurun_out/r6_count_nsub.err
  python - <<P
import json
d = json.loads(open("gpurun_out/r6_count_nsub${ns}.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("count CNMF_G2_NSUB=$ns:", round(d["value"], 1), "restarts/s; pass A/B TF", round(r["achieved_passA"]), round(r["achieved_passB"]), "frac", round(r["frac"], 3), "e2e", round(r["end_to_end"]["frac"], 3), "avg ms", r.get("avg_launch_ms"))
P
done
