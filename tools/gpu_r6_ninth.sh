#!/bin/bash
# Round-6 ninth GPU session: the emulated 8-GPU shard (113 restarts, all tail) at different starting widths.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
for kc in 1024 768 512 256; do
  CNMF_KC=$kc timeout 300 python bench.py --steps 2 --warmup 1 --emulate-rank 0/8 --no-cpu-baseline --no-extras > gpurun_out/r6_shard_kc.json 2>> gpurun_out/r6_shard_kc.err
  python - <<P
import json
d = json.loads(open("gpurun_out/r6_shard_kc.json").read().strip().splitlines()[-1])
c = d["config"]
print("shard 0/8 CNMF_KC=$kc:", round(d["value"], 1), "restarts/s of the shard; ms per step", round(d["ms_per_step"], 1), "outer iterations per step", c["tail"]["iterations_per_step"], "utilisation", round(c["column_utilisation"], 3), "kc", c["packed_columns"])
P
done 2>&1 | tee gpurun_out/r6_shard_width_ab.txt
