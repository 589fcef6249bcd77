#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/probe_gemm2h.py > gpurun_out/probe_gemm2h.log 2>&1; tail -32 gpurun_out/probe_gemm2h.log
timeout 600 python -m pytest tests/test_gpu_nmf.py tests/test_gpu_golden_big.py tests/test_gpu_configs.py -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1
grep -n "passed\|failed\|Error\|error\|assert" gpurun_out/pytest_gpu.log | tail -20
for v in 0 1 3; do
  CNMF_G2_VAR=$v timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2> gpurun_out/bench_v$v.err > gpurun_out/bench_v$v.json
  python - <<PY
import json
d=json.load(open("gpurun_out/bench_v$v.json"))
print("VAR $v: restarts/s %.1f  ms/step %.0f  passA %.4f passB %.4f ms  gemm_share %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"]["passA"], d["roofline"]["avg_launch_ms"]["passB"], d["roofline"]["gemm_share_of_gpu_time"]))
PY
done
bash tools/gpu_r2_prof.sh 2>&1 | tail -28
