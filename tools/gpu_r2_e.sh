#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_tail.py tests/test_gpu_consensus.py tests/test_gpu_pipeline.py -m gpu -q > gpurun_out/pytest_gpu.log 2>&1
grep -n "passed\|failed\|Error\|error\|assert\|^E " gpurun_out/pytest_gpu.log | tail -60
