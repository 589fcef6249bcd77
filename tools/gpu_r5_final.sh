#!/bin/bash
# Round-5 closing session on the final build: the whole GPU suite, smoke, the default bench line.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r5_pytest_final.log 2>&1; echo "pytest rc=$?" | tee gpurun_out/r5_final.status
tail -4 gpurun_out/r5_pytest_final.log
grep -a "C4 KL\|FAILED" gpurun_out/r5_pytest_final.log | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5_smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/r5_final.status
timeout 900 python bench.py > gpurun_out/r5_bench_default.json 2> gpurun_out/r5_bench_default.err; echo "bench rc=$?" | tee -a gpurun_out/r5_final.status
python - <<P
import json
d = json.loads(open("gpurun_out/r5_bench_default.json").read().strip().splitlines()[-1])
print("bench:", round(d["value"], 1), "restarts/s; roofline", round(d["roofline"]["frac"], 3), "passA/B TF", round(d["roofline"]["achieved_passA"]), round(d["roofline"]["achieved_passB"]), "e2e", round(d["roofline"]["end_to_end"]["frac"], 3), "gemm share", round(d["roofline"]["gemm_share_of_gpu_time"], 3), "tail", round(d["config"]["tail"]["share_of_gpu_time"], 3), "traffic", d["roofline"].get("traffic"))
print("hints:", d.get("with_queue_hints"))
print("general:", d["general_path"].get("restarts_per_s"), "consensus:", d["consensus"]["gpu_ms"], d["consensus"].get("gpu_ms_spectra_resident"))
print("kl:", json.dumps(d.get("kl_non_zero_path"))[:1200])
print("e2e:", d["e2e"]["stages_s"], d["e2e"]["total_s"])
print("cpu_baseline:", d["cpu_baseline"]["value"], d["cpu_baseline"]["unit"], d["cpu_baseline"].get("cores"))
print("ablation:", d["roofline"].get("mfma_only_ablation"))
P
