#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1
grep -n "passed\|failed\|Error\|error\|^E " gpurun_out/pytest_gpu.log | tail -30
for v in 0 1; do
  if [ $v = 1 ]; then export CNMF_NO_PSUM=1; else unset CNMF_NO_PSUM; fi
  timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2> gpurun_out/bench_p$v.err > gpurun_out/bench_p$v.json
  python - <<PY
import json
d=json.load(open("gpurun_out/bench_p$v.json"))
print("NO_PSUM=$v: restarts/s %.1f  ms/step %.0f  passA %.4f passB %.4f ms  gemm_share %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"]["passA"], d["roofline"]["avg_launch_ms"]["passB"], d["roofline"]["gemm_share_of_gpu_time"]))
PY
done
unset CNMF_NO_PSUM
bash tools/gpu_r2_prof.sh 2>&1 | grep -v "count_\|col_min\|fillBuffer\|copyBuffer\|rng_kernel" | tail -24
