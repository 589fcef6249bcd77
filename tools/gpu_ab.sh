#!/bin/bash
python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k soak 2>&1 | grep "passed\|failed\|rror\|^E" | tail -5
