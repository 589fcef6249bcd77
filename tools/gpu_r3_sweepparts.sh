#!/bin/bash
# A/B of the W/H sweep decomposition (workgroups per launch vs resident capacity) + a kernel trace of the default
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for p in 64 40 32 25 98 196; do
  CNMF_SWEEP_PARTS=$p timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r3_parts_$p.json 2> gpurun_out/r3_parts_$p.err
  python - <<PY
import json
d=json.load(open('gpurun_out/r3_parts_$p.json')); r=d['roofline']
print('parts $p: %.1f restarts/s  passA %.1f us passB %.1f us gemm share %.3f' % (d['value'], 1e3*r['avg_launch_ms']['passA'], 1e3*r['avg_launch_ms']['passB'], r['gemm_share_of_gpu_time']))
PY
done
PROF_TAG="round 3 start: LPT queue" bash tools/gpu_r2_prof.sh > gpurun_out/r3_prof0.log 2>&1
cp gpurun_out/kernel_stats.txt gpurun_out/r3_kernel_stats_start.txt
tail -14 gpurun_out/r3_prof0.log
