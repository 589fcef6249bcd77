#!/bin/bash
# Round-2 GPU session A: host facts, the GPU test suite, smoke.
mkdir -p gpurun_out
export TMPDIR=/tmp
( nproc; free -g | head -2; df -h /dev/shm | tail -1; rocm-smi --showmeminfo vram 2>/dev/null | head -8 ) > gpurun_out/host_facts.txt 2>&1
python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/pytest_gpu.log 2>&1
tail -40 gpurun_out/pytest_gpu.log
python __graft_entry__.py smoke 2>&1 | tail -2 | tee gpurun_out/smoke.log
cat gpurun_out/host_facts.txt
