#!/bin/bash
# The default bench line for the record (profiles/r6_bench_C3_default.json): `python bench.py` on the committed tree.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python bench.py > gpurun_out/r6_bench_default.json 2> gpurun_out/r6_bench_default.err; echo "bench rc=$?"
python - <<P
import json
d = json.loads(open("gpurun_out/r6_bench_default.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(round(d["value"], 1), "restarts/s; frac", round(r["frac"], 3), "traffic", r.get("traffic"), "e2e", round(r["end_to_end"]["frac"], 3), "general", round(d["general_path"]["restarts_per_s"], 1), "ablation stale:", (r.get("mfma_only_ablation") or {}).get("stale"))
P
