#!/bin/bash
# round 4, sixth GPU session: the GPU suite on the explicit queue-hint API, smoke, the full default bench line (extras:
# hinted step, general path, consensus, e2e), the shard projection
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4_pytest.log 2>&1; echo "pytest rc=$?" | tee gpurun_out/r4_final.status
tail -3 gpurun_out/r4_pytest.log
timeout 900 python bench.py > gpurun_out/r4_bench_default.json 2> gpurun_out/r4_bench_default.err; echo "bench rc=$?" | tee -a gpurun_out/r4_final.status
python - <<P
import json
d = json.loads(open("gpurun_out/r4_bench_default.json").read().strip().splitlines()[-1])
print("bench:", round(d["value"], 1), "restarts/s; roofline", round(d["roofline"]["frac"], 3), "e2e", round(d["roofline"]["end_to_end"]["frac"], 3), "tail", round(d["config"]["tail"]["share_of_gpu_time"], 3))
print("hints:", d.get("with_queue_hints"))
print("general:", d["general_path"].get("restarts_per_s"), "consensus:", d["consensus"]["gpu_ms"], d["consensus"].get("gpu_ms_spectra_resident"))
print("e2e:", d["e2e"]["stages_s"], d["e2e"]["total_s"])
P
timeout 600 python tools/shard_scaling.py --steps 2 --warmup 1 > gpurun_out/r4_shard.log 2>&1; echo "shard rc=$?" | tee -a gpurun_out/r4_final.status
tail -12 gpurun_out/r4_shard.log
