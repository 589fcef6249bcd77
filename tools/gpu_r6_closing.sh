#!/bin/bash
# Round-6 closing session ON THE COMMITTED TREE: the whole GPU suite, smoke, the kernel trace and the counters of the bench's OWN
# 900-restart step (count path and general path), the GEMM ablation probe, the other BASELINE configurations, the
# emulated 8-GPU shard, and the default bench line last.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
S=gpurun_out/r6_closing.status; : > $S
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r6_pytest_closing.log 2>&1; echo "pytest rc=$?" | tee -a $S
tail -4 gpurun_out/r6_pytest_closing.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6_smoke.log 2>&1; echo "smoke rc=$?" | tee -a $S
# kernel trace of the bench's own step (900 restarts)
PROF_TAG="default path, auto width" RPK=100 PROF_OUT=r6_kernel_stats_bench_C3.txt bash tools/gpu_prof.sh > gpurun_out/r6_prof.log 2>&1; echo "prof rc=$?" | tee -a $S; rm -rf gpurun_out/prof
head -16 gpurun_out/r6_kernel_stats_bench_C3.txt | cut -c1-90,111-170
# counters: count path, then the general path
PMC_OUT=r6_pmc_traffic_f16.json bash tools/gpu_pmc_bench.sh > gpurun_out/r6_pmc_f16.log 2>&1; echo "pmc f16 rc=$?" | tee -a $S
CNMF_NO_COUNTS=1 PMC_XPLANES=2 PMC_NOTE="general path (CNMF_NO_COUNTS=1: X as two f16 planes, gemm_mode 5, spread stream, identity XCD mapping)" PMC_OUT=r6_pmc_traffic_general.json bash tools/gpu_pmc_bench.sh > gpurun_out/r6_pmc_general.log 2>&1; echo "pmc general rc=$?" | tee -a $S
tail -30 gpurun_out/r6_pmc_general.log | head -40
# the ablation probe of the count GEMM at the production launch shape
ABLATE_SHAPES=2 timeout 300 python tools/probe_gemm2h_ablate.py > gpurun_out/r6_gemm2h_ablation.txt 2>&1; echo "ablate rc=$?" | tee -a $S
cat gpurun_out/r6_gemm2h_ablation.txt | tail -9
# the other BASELINE configurations and the emulated shard
timeout 300 python bench.py --steps 5 --warmup 2 --workload C2 --kmin 10 --kmax 10 --no-cpu-baseline --no-extras > gpurun_out/r6_cfg_c2.json 2> gpurun_out/r6_cfg_c2.err; echo "c2 rc=$?" | tee -a $S
timeout 600 python bench.py --steps 2 --warmup 1 --workload C4 --kmin 20 --kmax 20 --no-cpu-baseline --no-extras > gpurun_out/r6_cfg_c4.json 2> gpurun_out/r6_cfg_c4.err; echo "c4 rc=$?" | tee -a $S
timeout 600 python bench.py --steps 1 --warmup 1 --restarts-per-k 200 --no-cpu-baseline --no-extras > gpurun_out/r6_cfg_c3_1800.json 2> gpurun_out/r6_cfg_c3_1800.err; echo "c3x200 rc=$?" | tee -a $S
timeout 600 python bench.py --steps 2 --warmup 1 --emulate-rank 0/8 --no-cpu-baseline --no-extras > gpurun_out/r6_shard8.json 2> gpurun_out/r6_shard8.err; echo "shard rc=$?" | tee -a $S
for f in c2 c4 c3_1800; do python - <<P
import json
d = json.loads(open("gpurun_out/r6_cfg_$f.json").read().strip().splitlines()[-1])
print("$f:", round(d["value"], 1), "restarts/s,", round(d["ms_per_step"], 2), "ms per job, kc", d["config"]["packed_columns"], "frac", round(d["roofline"]["frac"], 3), "e2e", round(d["roofline"]["end_to_end"]["frac"], 3), "gemm share", round(d["roofline"]["gemm_share_of_gpu_time"], 3))
P
done
python - <<P
import json
d = json.loads(open("gpurun_out/r6_shard8.json").read().strip().splitlines()[-1])
print("shard 0/8:", round(d["value"], 1), "restarts/s of the shard; tail share", round(d["config"]["tail"]["share_of_gpu_time"], 3))
P
# the default bench line, last
timeout 900 python bench.py > gpurun_out/r6_bench_default.json 2> gpurun_out/r6_bench_default.err; echo "bench rc=$?" | tee -a $S
python - <<P
import json
d = json.loads(open("gpurun_out/r6_bench_default.json").read().strip().splitlines()[-1])
print("bench:", round(d["value"], 1), "restarts/s; roofline", round(d["roofline"]["frac"], 3), "passA/B TF", round(d["roofline"]["achieved_passA"]), round(d["roofline"]["achieved_passB"]), "e2e", round(d["roofline"]["end_to_end"]["frac"], 3), "gemm share", round(d["roofline"]["gemm_share_of_gpu_time"], 3), "tail", round(d["config"]["tail"]["share_of_gpu_time"], 3), "traffic", d["roofline"].get("traffic"))
print("regime:", d["config"].get("regime"))
print("hints:", d.get("with_queue_hints"))
print("general:", {k: d["general_path"].get(k) for k in ("restarts_per_s", "frac", "avg_launch_ms")}, "consensus:", d["consensus"]["gpu_ms"], d["consensus"].get("gpu_ms_spectra_resident"))
print("kl:", json.dumps(d.get("kl_non_zero_path"))[:900])
print("e2e:", d["e2e"]["stages_s"], d["e2e"]["total_s"], d["e2e"].get("consensus_spectra_sumsq_vs_cpu"))
print("cpu_baseline:", d["cpu_baseline"]["value"], d["cpu_baseline"]["unit"], d["cpu_baseline"].get("cores"), d["cpu_baseline"].get("full_restarts_offline"))
print("ablation:", d["roofline"].get("mfma_only_ablation"))
P
cat $S
