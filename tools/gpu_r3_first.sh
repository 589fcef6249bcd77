#!/bin/bash
# round 3, first GPU call: the whole GPU suite, the default bench line, the queue-order A/B, the shard projection
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_pytest.log 2>&1; echo "pytest rc=$?" > gpurun_out/r3_first.status
timeout 600 python bench.py > gpurun_out/r3_bench_default.json 2> gpurun_out/r3_bench_default.err; echo "bench rc=$?" >> gpurun_out/r3_first.status
CNMF_QUEUE=rank timeout 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r3_bench_queue_rank.json 2> gpurun_out/r3_bench_queue_rank.err; echo "bench(rank queue) rc=$?" >> gpurun_out/r3_first.status
timeout 600 python tools/shard_scaling.py --steps 1 --warmup 1 > gpurun_out/r3_shard.log 2>&1; echo "shard rc=$?" >> gpurun_out/r3_first.status
CNMF_QUEUE=rank timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 1 --emulate-rank 0/8 > gpurun_out/r3_bench_shard8_queue_rank.json 2>/dev/null
cat gpurun_out/r3_first.status; tail -5 gpurun_out/r3_pytest.log
