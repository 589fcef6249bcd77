#!/bin/bash
# Round-5 fourth GPU session: the whole GPU suite on the round-5 library, smoke, kernel trace + PMC traffic of the bench step,
# shard projection.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5_pytest.log 2>&1; echo "pytest rc=$?" | tee gpurun_out/r5_fourth.status
tail -4 gpurun_out/r5_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5_smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/r5_fourth.status
PROF_TAG="default path, auto width" RPK=50 PROF_OUT=r5_kernel_stats.txt bash tools/gpu_prof.sh > gpurun_out/r5_prof.log 2>&1; rm -rf gpurun_out/prof
head -12 gpurun_out/r5_kernel_stats.txt | cut -c1-90,111-170
RPK=50 PMC_OUT=r5_pmc_traffic.json bash tools/gpu_pmc_bench.sh > gpurun_out/r5_pmc.log 2>&1
tail -32 gpurun_out/r5_pmc.log | head -28
timeout 600 python tools/shard_scaling.py --steps 2 --warmup 1 > gpurun_out/r5_shard.log 2>&1; echo "shard rc=$?" | tee -a gpurun_out/r5_fourth.status
grep -n "projected_efficiency\|world" gpurun_out/r5_shard.log | tail -8
