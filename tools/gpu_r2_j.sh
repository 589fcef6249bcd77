#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_nmf.py tests/test_gpu_golden_big.py tests/test_gpu_configs.py tests/test_gpu_consensus.py tests/test_gpu_tail.py -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1
grep -n "passed\|failed\|Error\|error\|^E " gpurun_out/pytest_gpu.log | tail -12
bash tools/gpu_r2_prof.sh 2>&1 | grep -v "count_\|col_min\|fillBuffer\|copyBuffer\|rng_kernel" | tail -22
/usr/bin/time -v timeout 900 python bench.py 2> gpurun_out/bench.err > gpurun_out/bench.json; grep -E "Elapsed|Maximum resident" gpurun_out/bench.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/bench.json"))
print("restarts/s %.1f ms/step %.0f passA %.4f passB %.4f frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"]["passA"], d["roofline"]["avg_launch_ms"]["passB"], d["roofline"]["frac"]))
print(json.dumps(d["cpu_baseline"], indent=0)[:1500])
print(d.get("consensus"))
PY
