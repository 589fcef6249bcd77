#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
for lag in 2 4 6; do
  CNMF_LAG=$lag timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2> gpurun_out/bench_l$lag.err > gpurun_out/bench_l$lag.json
  python - <<PY
import json
d=json.load(open("gpurun_out/bench_l$lag.json"))
print("LAG $lag: restarts/s %.1f  ms/step %.0f  passA %.4f passB %.4f ms  gemm_share %.3f util %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"]["passA"], d["roofline"]["avg_launch_ms"]["passB"], d["roofline"]["gemm_share_of_gpu_time"], d["config"]["column_utilisation"]))
PY
done
timeout 900 python bench.py 2> gpurun_out/bench.err > gpurun_out/bench.json; python - <<'PY'
import json
d=json.load(open("gpurun_out/bench.json"))
print("restarts/s %.1f ms/step %.0f passA %.4f passB %.4f frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"]["passA"], d["roofline"]["avg_launch_ms"]["passB"], d["roofline"]["frac"]))
print(json.dumps(d["cpu_baseline"], indent=0)[:1800])
print(d.get("consensus"))
PY
tail -3 gpurun_out/bench.err
