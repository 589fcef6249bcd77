#!/bin/bash
python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_consensus.py -m gpu -x -q 2>&1 | grep "passed\|failed\|rror" | tail -3
python tools/e2e_c3.py > gpurun_out/e2e_c3.json 2> gpurun_out/e2e.err; head -c 800 gpurun_out/e2e_c3.json; echo
