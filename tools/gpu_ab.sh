#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1
tail -3 gpurun_out/gpu_tests.log
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
wc -l gpurun_out/bench_default.json
