#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
AB_CELLS=14000 AB_RESTARTS=120 AB_ITERS=12 AB_W=1 timeout 600 python tools/fused_ab.py 2>&1 | tail -25
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_pipeline.py tests/test_gpu_tail.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -4
