#!/bin/bash
# Round-4 measurement session, second part (W sweep back at 5 waves per SIMD): kernel trace, default bench line, PMC traffic
# (count path and general path), MFMA-only ablation of the GEMM stream
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
PROF_TAG="default path, auto width" RPK=50 PROF_OUT=r4_kernel_stats.txt bash tools/gpu_r3_prof.sh > gpurun_out/r4_prof.log 2>&1; rm -rf gpurun_out/prof
head -10 gpurun_out/r4_kernel_stats.txt | cut -c1-90,111-170
timeout 600 python -m pytest tests/test_gpu_nmf.py tests/test_gpu_golden_big.py tests/test_gpu_determinism.py -m gpu -x -q 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r4_bench_default.json 2> gpurun_out/r4_bench_default.err; echo "bench rc=$?"
python - <<P
import json
d = json.loads(open("gpurun_out/r4_bench_default.json").read().strip().splitlines()[-1])
print("bench:", round(d["value"], 1), "restarts/s; roofline", round(d["roofline"]["frac"], 3), "passA/B TF", round(d["roofline"]["achieved_passA"]), round(d["roofline"]["achieved_passB"]), "e2e", round(d["roofline"]["end_to_end"]["frac"], 3), "gemm share", round(d["roofline"]["gemm_share_of_gpu_time"], 3), "tail", round(d["config"]["tail"]["share_of_gpu_time"], 3))
print("hints:", d.get("with_queue_hints", {}).get("restarts_per_s"), d.get("with_queue_hints", {}).get("tail_share_of_gpu_time"))
print("general:", d["general_path"].get("restarts_per_s"), "consensus:", d["consensus"]["gpu_ms"], d["consensus"].get("gpu_ms_spectra_resident"))
print("e2e:", d["e2e"]["stages_s"], d["e2e"]["total_s"])
P
RPK=50 bash tools/gpu_pmc_bench.sh > gpurun_out/r4_pmc.log 2>&1
cp gpurun_out/pmc_traffic.json gpurun_out/r4_pmc_traffic_f16.json
CNMF_NO_COUNTS=1 PMC_XPLANES=2 PMC_OUT=r4_pmc_traffic_general.json PMC_NOTE="count detection off (CNMF_NO_COUNTS=1): X as two f16 planes, three plane pairs per product (gemm_mode 5)" RPK=20 bash tools/gpu_pmc_bench.sh > gpurun_out/r4_pmc_general.log 2>&1
tail -22 gpurun_out/r4_pmc_general.log | head -14
timeout 300 python tools/probe_gemm2h_ablate.py > gpurun_out/r4_ablate.txt 2>&1
python - <<PY
import json, re, sys
sys.path.insert(0, '.')
from bench import source_hashes
vals = {}
for ln in open('gpurun_out/r4_ablate.txt'):
    m = re.match(r'(pass\S+).* var=(\d+): ([\d.]+) ms \((\d+) TF', ln)
    if m: vals.setdefault((m.group(1), int(m.group(2))), []).append(float(m.group(4)))
pb = {v: max(x) for (t, v), x in vals.items() if t.startswith('passB')}
json.dump({"_source": "tools/probe_gemm2h_ablate.py (pass-B shape 256 x 50176 x 2048, nsub 2, 20 repetitions, best of 2): f16 MFMA TF issued by the production stream (var 4) and by the same stream with everything but its MFMAs removed (var 7)",
           "production_tflops_issued": pb.get(4), "mfma_only_tflops_issued": pb.get(7), "without_dma": pb.get(3), "mfma_and_barriers": pb.get(6),
           "kernel_source_sha256": source_hashes()}, open('gpurun_out/r4_gemm2h_ablation.json', 'w'), indent=1)
print(open('gpurun_out/r4_gemm2h_ablation.json').read())
PY
