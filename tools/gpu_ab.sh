#!/bin/bash
export CNMF_BENCH_BACKEND=gloo CNMF_BENCH_ONE_GPU=1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 2 --steps 1 --warmup 1 --restarts-per-k 5 > gpurun_out/bench_2rank.json 2> gpurun_out/bench_2rank.err
echo "rc=$?"; cut -c1-330 gpurun_out/bench_2rank.json; tail -3 gpurun_out/bench_2rank.err | cut -c1-200
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_2rank.json').read().strip().split('\n')[-1])
print(d['n_gpus'], d['value'], d['config']['restarts_per_step_per_gpu'], d['config']['parallelism'], d['config']['gather'])
PY
