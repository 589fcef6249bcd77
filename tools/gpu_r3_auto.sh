#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_nmf.py tests/test_gpu_golden_big.py tests/test_gpu_configs.py tests/test_gpu_pipeline.py tests/test_gpu_determinism.py -m gpu -x -q > gpurun_out/r3_auto_pytest.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r3_auto_pytest.log
timeout 600 python tools/shard_scaling.py --steps 2 --warmup 1 > gpurun_out/r3_shard.log 2>&1; echo "shard rc=$?"
tail -22 gpurun_out/r3_shard.log
