#!/bin/bash
python -m pytest tests/test_gpu_comm.py -m gpu -x -q 2>&1 | grep "passed\|failed\|rror\|^E" | tail -5
