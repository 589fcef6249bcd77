#!/bin/bash
# Round-6 fifth GPU session: BASELINE config 4 (and 2) with the round-5 library against the round-6 one on ONE box, alternating
# (the closing rehearsal measured C4 at 391 ms per job against round 5's 338 on another box with identical kernel times).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
  for lib in tools/bin/libcnmf_r5.so ""; do
    CNMF_LIB_PATH=${lib:+$GRAFT_REPO_ROOT/$lib} timeout 600 python bench.py --steps 3 --warmup 1 --workload C4 --kmin 20 --kmax 20 --no-cpu-baseline --no-extras > gpurun_out/r6_c4_ab.json 2> gpurun_out/r6_c4_ab.err
    python - <<P
import json
d = json.loads(open("gpurun_out/r6_c4_ab.json").read().strip().splitlines()[-1])
r = d["config"]["per_rank"][0]
print("C4 lib=${lib:-round 6 (default)} rep $rep:", round(d["ms_per_step"], 1), "ms per job; gpu ms per step", round(r["gpu_ms"] / d["steps"], 1), "-> host", round(d["ms_per_step"] - r["gpu_ms"] / d["steps"], 1))
P
  done
done 2>&1 | tee gpurun_out/r6_c4_lib_ab.txt
