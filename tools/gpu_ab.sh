#!/bin/bash
python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q -k normalisation 2>&1 | grep -v "RCCL\|HIP v\|ROCm\|Hostname\|Librccl" | tail -25
