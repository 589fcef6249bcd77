#!/bin/bash
# kernel traces of the two-engine probe: one engine vs two engines on one GPU
mkdir -p gpurun_out; export TMPDIR=/tmp
for m in one two; do
  rm -rf gpurun_out/ov_$m
  MODE=$m REPS=1 PER_K=16 timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/ov_$m -o trace -- python tools/probe_two_engines.py > gpurun_out/ov_$m.log 2>&1
  grep "engine" gpurun_out/ov_$m.log
  db=$(find gpurun_out/ov_$m -name "*.db" | head -1)
  python tools/overlap_report.py $db 60 2>&1 | tail -80
done
