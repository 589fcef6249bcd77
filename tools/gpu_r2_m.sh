#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_nmf.py tests/test_gpu_golden_big.py tests/test_gpu_configs.py tests/test_gpu_pipeline.py tests/test_gpu_edges.py -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1
grep -n "passed\|failed\|Error\|error\|^E " gpurun_out/pytest_gpu.log | tail -12
bash tools/gpu_r2_prof.sh 2>&1 | grep -v "count_\|col_min\|fillBuffer\|copyBuffer\|rng_kernel" | tail -12
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2> gpurun_out/bench_q.err > gpurun_out/bench_q.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_q.json"))
print("restarts/s %.1f  ms/step %.0f  passA %.4f passB %.4f ms  gemm_share %.3f util %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"]["passA"], d["roofline"]["avg_launch_ms"]["passB"], d["roofline"]["gemm_share_of_gpu_time"], d["config"]["column_utilisation"]))
PY
