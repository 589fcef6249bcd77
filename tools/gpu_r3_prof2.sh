#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof
( cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o trace -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --emulate-rank 0/8 > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1 )
DB=$(ls gpurun_out/prof/*/*results.db gpurun_out/prof/*results.db 2>/dev/null | head -1)
python tools/export_profile.py $DB gpurun_out/r3_kernel_stats_shard8.txt "shard 0/8" | head -16 | cut -c1-100,111-160
python - <<PY
import sqlite3
c=sqlite3.connect("$DB")
rows=list(c.execute("select name,start,end from kernels order by start"))
# per-kernel duration over time for the two GEMM kernels: print every 100th launch
import collections
k=collections.defaultdict(list)
for n,s,e in rows:
    if 'gemm2h' in n: k[n.split('(')[0][-60:]].append((e-s)/1e3)
for n,v in k.items():
    print(n, len(v), ' '.join('%.0f'%x for x in v[::max(1,len(v)//24)]))
PY
rm -rf gpurun_out/prof
