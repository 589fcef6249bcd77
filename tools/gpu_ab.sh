#!/bin/bash
python -m pytest tests/test_hvg.py tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | grep "passed\|failed\|rror\|^E" | tail -5
