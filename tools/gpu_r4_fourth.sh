#!/bin/bash
# round 4, fourth GPU session: the whole GPU suite, then the default bench line WITH extras (general path on 3 MFMAs,
# consensus with resident spectra, e2e) and the general path with the fourth plane pair back (CNMF_G2_GEN4=1)
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/r4_gpu_tests.log
tail -12 gpurun_out/r4_gpu_tests.log
timeout 900 python bench.py --steps 4 --warmup 1 > gpurun_out/r4_bench_full.json 2> gpurun_out/r4_bench_full.err
python - <<P
import json
try:
    d = json.loads(open("gpurun_out/r4_bench_full.json").read().strip().splitlines()[-1])
    print("bench:", round(d["value"], 1), "restarts/s", round(d["ms_per_step"]), "ms; roofline", d["roofline"]["frac"], "e2e frac", d["roofline"].get("end_to_end"))
    print("general:", {k: d["general_path"].get(k) for k in ("value", "restarts_per_s", "achieved_TFLOPs", "frac", "peak_TFLOPs")})
    print("consensus:", d["consensus"])
    print("e2e:", d["e2e"]["stages_s"], d["e2e"]["total_s"], d["e2e"]["cpu_reference"]["stages_s"], d["e2e"].get("consensus_spectra_sumsq_vs_cpu"))
    print("cpu_baseline:", d["cpu_baseline"]["value"], d["cpu_baseline"]["unit"])
except Exception as e:
    print("bench failed", e); print(open("gpurun_out/r4_bench_full.err").read()[-2500:])
P
CNMF_G2_GEN4=1 timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r4_bench_gen4.json 2> gpurun_out/r4_bench_gen4.err
python - <<P
import json
try:
    d = json.loads(open("gpurun_out/r4_bench_gen4.json").read().strip().splitlines()[-1])
    print("general, 4 MFMAs:", {k: d["general_path"].get(k) for k in ("value", "restarts_per_s", "achieved_TFLOPs", "frac", "peak_TFLOPs")})
except Exception as e:
    print("gen4 failed", e); print(open("gpurun_out/r4_bench_gen4.err").read()[-1500:])
P
