#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
CNMF_DEBUG=1 timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > gpurun_out/r4_fcdbg.json 2> gpurun_out/r4_fcdbg.err
grep "expected per rank" gpurun_out/r4_fcdbg.err | head -40 | cut -c1-400
python tools/dump_iters.py 2>/dev/null | tail -3
