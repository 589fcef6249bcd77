#!/bin/bash
python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k C4 2>&1 | grep -v "RCCL\|HIP v\|ROCm\|Hostname\|Librccl" | tail -15
