#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_comm.py -m gpu -x -q > gpurun_out/comm_test.log 2>&1
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 CNMF_BENCH_FORCE_DIST=1
for g in torch rccl; do
  CNMF_GATHER=$g timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --restarts-per-k 10 2>gpurun_out/err_$g.log >gpurun_out/out_$g.log
done
