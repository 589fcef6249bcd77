#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_nmf.py tests/test_gpu_golden_big.py tests/test_gpu_configs.py tests/test_gpu_edges.py -m gpu -x -q > gpurun_out/r3_repack_pytest.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r3_repack_pytest.log
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r3_repack_bench.json 2> gpurun_out/r3_repack_bench.err
python - <<PY
import json
d=json.load(open('gpurun_out/r3_repack_bench.json')); r=d['roofline']; c=d['config']
print('%.1f restarts/s  passA %.1f us passB %.1f us gemm share %.3f util %.3f tail %.0f ms kc %d' % (d['value'], 1e3*r['avg_launch_ms']['passA'], 1e3*r['avg_launch_ms']['passB'], r['gemm_share_of_gpu_time'], c['column_utilisation'], c['tail']['ms_per_step'], c['packed_columns']))
PY
bash tools/gpu_r3_prof.sh
