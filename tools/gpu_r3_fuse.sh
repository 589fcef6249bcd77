#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_nmf.py tests/test_gpu_golden_big.py tests/test_gpu_configs.py tests/test_gpu_determinism.py tests/test_gpu_edges.py -m gpu -x -q > gpurun_out/r3_fuse_pytest.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r3_fuse_pytest.log
for f in 1 0; do
  CNMF_FUSE_W=$f timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('fuse=$f: %.1f restarts/s  passA %.1f passB %.1f us  gemm share %.3f e2e frac %.3f' % (d['value'], 1e3*r['avg_launch_ms']['passA'], 1e3*r['avg_launch_ms']['passB'], r['gemm_share_of_gpu_time'], r['end_to_end']['frac']))"
done
