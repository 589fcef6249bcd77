#!/bin/bash
# final validation of the round-4 tree: the whole GPU suite, smoke, the default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r4_pytest.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r4_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r4_smoke.log
timeout 900 python bench.py > gpurun_out/r4_bench_last.json 2> gpurun_out/r4_bench_last.err; echo "bench rc=$?"
python - <<P
import json
d = json.loads(open("gpurun_out/r4_bench_last.json").read().strip().splitlines()[-1])
print("bench:", round(d["value"], 1), "restarts/s; roofline", round(d["roofline"]["frac"], 3), "traffic", d["roofline"].get("traffic"), "e2e", round(d["roofline"]["end_to_end"]["frac"], 3), "tail", round(d["config"]["tail"]["share_of_gpu_time"], 3))
print("hints:", d.get("with_queue_hints", {}).get("restarts_per_s"), "general:", d["general_path"].get("restarts_per_s"), "consensus:", d["consensus"]["gpu_ms"], d["consensus"].get("gpu_ms_spectra_resident"), "e2e:", d["e2e"]["total_s"])
print("mfma ablation:", d["roofline"].get("mfma_only_ablation"))
P
