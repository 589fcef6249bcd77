#!/bin/bash
# round 4, fifth GPU session: the GPU suite (determinism test re-stated, finalize latency fix), smoke, the default bench
# line, the kernel trace of a 450-restart step, PMC traffic of the bench step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4_pytest.log 2>&1; echo "pytest rc=$?" | tee gpurun_out/r4_final.status
tail -4 gpurun_out/r4_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4_smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/r4_final.status
timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r4_bench_noextras.json 2> gpurun_out/r4_bench_noextras.err; echo "bench rc=$?" | tee -a gpurun_out/r4_final.status
python - <<P
import json
d = json.loads(open("gpurun_out/r4_bench_noextras.json").read().strip().splitlines()[-1])
print("bench:", round(d["value"], 1), "restarts/s", round(d["ms_per_step"]), "ms; roofline", round(d["roofline"]["frac"], 3), "e2e", round(d["roofline"]["end_to_end"]["frac"], 3), "gemm share", round(d["roofline"]["gemm_share_of_gpu_time"], 3), "tail", round(d["config"]["tail"]["share_of_gpu_time"], 3))
P
PROF_TAG="default path, auto width" RPK=50 PROF_OUT=r4_kernel_stats.txt bash tools/gpu_r3_prof.sh > gpurun_out/r4_prof.log 2>&1; rm -rf gpurun_out/prof
head -14 gpurun_out/r4_kernel_stats.txt 2>/dev/null | cut -c1-90,111-170 || tail -5 gpurun_out/r4_prof.log
RPK=50 bash tools/gpu_pmc_bench.sh > gpurun_out/r4_pmc.log 2>&1
tail -30 gpurun_out/r4_pmc.log | head -40
