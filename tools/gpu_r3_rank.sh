#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_edges.py tests/test_gpu_consensus.py tests/test_gpu_tail.py tests/test_gpu_pipeline.py -m gpu -x -q > gpurun_out/r3_rank_pytest.log 2>&1; echo "pytest rc=$?"
tail -25 gpurun_out/r3_rank_pytest.log
