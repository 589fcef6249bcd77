#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_nmf.py tests/test_gpu_golden_big.py tests/test_gpu_configs.py tests/test_gpu_edges.py tests/test_gpu_pipeline.py -m gpu -x -q > gpurun_out/r3_wide_pytest.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/r3_wide_pytest.log
for kc in 256 512 768 1024; do
  CNMF_KC=$kc timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r3_kc_$kc.json 2> gpurun_out/r3_kc_$kc.err
  python - <<PY
import json
d=json.load(open('gpurun_out/r3_kc_$kc.json')); r=d['roofline']; c=d['config']
print('KC $kc: %.1f restarts/s  passA %.1f us passB %.1f us gemm share %.3f util %.3f tail %.0f ms kc %d' % (d['value'], 1e3*r['avg_launch_ms']['passA'], 1e3*r['avg_launch_ms']['passB'], r['gemm_share_of_gpu_time'], c['column_utilisation'], c['tail']['ms_per_step'], c['packed_columns']))
PY
done
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r3_kc_auto.json 2> gpurun_out/r3_kc_auto.err
python -c "
import json
d=json.load(open('gpurun_out/r3_kc_auto.json')); print('auto: %.1f restarts/s kc %d' % (d['value'], d['config']['packed_columns']))"
