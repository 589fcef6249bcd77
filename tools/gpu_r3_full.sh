#!/bin/bash
# full GPU suite + default bench line + NNDSVD probe
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r3_pytest.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r3_pytest.log
timeout 900 python bench.py > gpurun_out/r3_bench_default.json 2> gpurun_out/r3_bench_default.err; echo "bench rc=$?"
timeout 300 python tools/nndsvd_probe.py > gpurun_out/r3_nndsvd.log 2>&1; tail -14 gpurun_out/r3_nndsvd.log
