#!/bin/bash
# Round-2 final GPU session: full GPU test suite, smoke, default bench (with cpu_baseline + consensus), kernel trace, consensus trace
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --durations=6 > gpurun_out/pytest_gpu.log 2>&1
grep -n "passed\|failed\|Error\|error\|^E " gpurun_out/pytest_gpu.log | tail -12
python __graft_entry__.py smoke 2>&1 | tail -2 | tee gpurun_out/smoke.log
SECONDS=0
timeout 900 python bench.py 2> gpurun_out/bench.err > gpurun_out/bench.json; echo "bench wall ${SECONDS}s"; python - <<'PY'
import json
d=json.load(open("gpurun_out/bench.json"))
print("restarts/s %.1f ms/step %.0f passA %.4f passB %.4f frac %.3f hbm %s" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"]["passA"], d["roofline"]["avg_launch_ms"]["passB"], d["roofline"]["frac"], d["roofline"].get("hbm")))
cb=d["cpu_baseline"]; print(cb["value"], cb["mode"], cb["modes"]["workers_single_thread"], cb["modes"]["one_worker_all_threads"])
print(d.get("consensus"))
PY
tail -2 gpurun_out/bench.err
bash tools/gpu_r2_prof.sh 2>&1 | grep -v "count_\|col_min\|fillBuffer\|copyBuffer\|rng_kernel" | tail -8
# multiplicative-update solver: probe, kernel trace, PMC
bash tools/gpu_r2_mu_final.sh > gpurun_out/mu_final.log 2>&1; grep "KL k" gpurun_out/mu_final.log | head -9
# end to end, north-star job with the TPM tail; and the same job under beta_loss = kullback-leibler at n_iter = 10
timeout 600 python tools/e2e_c3.py 2>/dev/null | tail -1 | cut -c1-600
BETA_LOSS=kullback-leibler N_ITER=10 timeout 600 python tools/e2e_c3.py 2>/dev/null | tail -1 | cut -c1-600
