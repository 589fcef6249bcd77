#!/bin/bash
# Consensus latency session: parity tests that touch the consensus core, then C5 wall-clock with per-stage laps and a kernel trace.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_consensus.py tests/test_gpu_tail.py tests/test_gpu_option_b_replay.py tests/test_gpu_edges.py tests/test_gpu_pipeline.py -m gpu -x -q > gpurun_out/r3_cons_pytest.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r3_cons_pytest.log
CPU=0 timeout 300 python tools/gpu_cons.py > gpurun_out/r3_cons_time.log 2>&1
CPU=0 CNMF_DEBUG=1 timeout 300 python tools/gpu_cons.py > gpurun_out/r3_cons_laps.log 2>&1
cat gpurun_out/r3_cons_time.log; tail -12 gpurun_out/r3_cons_laps.log
cd /tmp && CPU=0 timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_cons -o cons -- python $GRAFT_REPO_ROOT/tools/gpu_cons.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
DB=$(ls gpurun_out/prof_cons/*/*results.db gpurun_out/prof_cons/*results.db 2>/dev/null | head -1)
python tools/export_profile.py $DB gpurun_out/r3_cons_kernels.txt "tools/gpu_cons.py: 3 x consensus core on C5 (5000 spectra x 2000 genes, k=20) + 1 x stats mode" > /dev/null 2>&1
rm -rf gpurun_out/prof_cons
head -40 gpurun_out/r3_cons_kernels.txt
