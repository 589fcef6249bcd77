#!/bin/bash
# Round-3 measurement session on one MI355X: GPU suite, smoke, default bench line, kernel trace, PMC traffic, MFMA-only
# ablation, shard projection, NNDSVD probe.  Everything lands in gpurun_out/ (copy + stamp into profiles/ afterwards).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3_pytest.log 2>&1; echo "pytest rc=$?" | tee gpurun_out/r3_final.status
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3_smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/r3_final.status
timeout 900 python bench.py > gpurun_out/r3_bench_default.json 2> gpurun_out/r3_bench_default.err; echo "bench rc=$?" | tee -a gpurun_out/r3_final.status
PROF_TAG="default path, auto width" RPK=50 bash tools/gpu_r3_prof.sh > gpurun_out/r3_prof.log 2>&1; rm -rf gpurun_out/prof
bash tools/gpu_pmc_bench.sh > gpurun_out/r3_pmc.log 2>&1
timeout 300 python tools/probe_gemm2h_ablate.py > gpurun_out/r3_ablate.txt 2>&1
python - <<PY
import json, re, sys
sys.path.insert(0, '.')
from bench import source_hashes
vals = {}
for ln in open('gpurun_out/r3_ablate.txt'):
    m = re.match(r'(pass\S+).* var=(\d+): ([\d.]+) ms \((\d+) TF', ln)
    if m: vals.setdefault((m.group(1), int(m.group(2))), []).append(float(m.group(4)))
pb = {v: max(x) for (t, v), x in vals.items() if t.startswith('passB')}
json.dump({"_source": "tools/probe_gemm2h_ablate.py (pass-B shape 256 x 50176 x 2048, nsub 2, 20 repetitions, best of 2): f16 MFMA TF issued by the production stream (var 4) and by the same stream with everything but its MFMAs removed (var 7)",
           "production_tflops_issued": pb.get(4), "mfma_only_tflops_issued": pb.get(7), "without_dma": pb.get(3), "mfma_and_barriers": pb.get(6),
           "kernel_source_sha256": source_hashes()}, open('gpurun_out/gemm2h_ablation.json', 'w'), indent=1)
print(open('gpurun_out/gemm2h_ablation.json').read())
PY
timeout 600 python tools/shard_scaling.py --steps 2 --warmup 1 > gpurun_out/r3_shard.log 2>&1; echo "shard rc=$?" | tee -a gpurun_out/r3_final.status
timeout 300 python tools/nndsvd_probe.py > gpurun_out/r3_nndsvd.log 2>&1
cat gpurun_out/r3_final.status; tail -2 gpurun_out/r3_pytest.log | head -1
