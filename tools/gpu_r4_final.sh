#!/bin/bash
# Round-4 measurement session on one MI355X: GPU suite, smoke, default bench line (all extras), kernel trace of a 450-restart
# step, PMC traffic of the bench step, shard projection.  Everything lands in gpurun_out/ (copy + stamp into profiles/).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r4_pytest.log 2>&1; echo "pytest rc=$?" | tee gpurun_out/r4_final.status
tail -3 gpurun_out/r4_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4_smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/r4_final.status
timeout 900 python bench.py > gpurun_out/r4_bench_default.json 2> gpurun_out/r4_bench_default.err; echo "bench rc=$?" | tee -a gpurun_out/r4_final.status
python - <<P
import json
d = json.loads(open("gpurun_out/r4_bench_default.json").read().strip().splitlines()[-1])
print("bench:", round(d["value"], 1), "restarts/s; roofline", round(d["roofline"]["frac"], 3), "passA/B TF", round(d["roofline"]["achieved_passA"]), round(d["roofline"]["achieved_passB"]), "e2e", round(d["roofline"]["end_to_end"]["frac"], 3), "gemm share", round(d["roofline"]["gemm_share_of_gpu_time"], 3), "tail", round(d["config"]["tail"]["share_of_gpu_time"], 3))
print("hints:", d.get("with_queue_hints"))
print("general:", d["general_path"].get("restarts_per_s"), "consensus:", d["consensus"]["gpu_ms"], d["consensus"].get("gpu_ms_spectra_resident"))
print("e2e:", d["e2e"]["stages_s"], d["e2e"]["total_s"], d["e2e"]["cpu_reference"]["stages_s"])
print("cpu_baseline:", d["cpu_baseline"]["value"], d["cpu_baseline"]["unit"], d["cpu_baseline"].get("cores"))
P
PROF_TAG="default path, auto width" RPK=50 PROF_OUT=r4_kernel_stats.txt bash tools/gpu_prof.sh > gpurun_out/r4_prof.log 2>&1; rm -rf gpurun_out/prof
head -12 gpurun_out/r4_kernel_stats.txt | cut -c1-90,111-170
RPK=50 bash tools/gpu_pmc_bench.sh > gpurun_out/r4_pmc.log 2>&1
tail -32 gpurun_out/r4_pmc.log | head -28
timeout 600 python tools/shard_scaling.py --steps 2 --warmup 1 > gpurun_out/r4_shard.log 2>&1; echo "shard rc=$?" | tee -a gpurun_out/r4_final.status
grep -n "projected_efficiency\|world" gpurun_out/r4_shard.log | tail -8
