#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for cfg in "1 0" "1 1" "0 0"; do
  set -- $cfg; xm=$1; nt=$2
  export CNMF_G2_XMAP=$xm
  if [ $nt = 1 ]; then export CNMF_G2_NT=1; else unset CNMF_G2_NT; fi
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r3_xm${xm}_nt$nt.json 2> gpurun_out/r3_xm${xm}_nt$nt.err
  python - <<PY
import json
d=json.load(open('gpurun_out/r3_xm${xm}_nt$nt.json')); r=d['roofline']; c=d['config']
print('xmap=$xm nt=$nt: %.1f restarts/s  passA %.1f us passB %.1f us gemm share %.3f' % (d['value'], 1e3*r['avg_launch_ms']['passA'], 1e3*r['avg_launch_ms']['passB'], r['gemm_share_of_gpu_time']))
PY
  rm -rf /tmp/pmcnt; ( cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmcnt -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --restarts-per-k 30 --no-cpu-baseline --no-extras > /dev/null 2>&1 )
  python - <<PY
import csv, glob
v=[]
for f in glob.glob('/tmp/pmcnt/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gemm2h_streamk' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE': v.append(float(r['Counter_Value']))
v.sort(); print('   pass A FETCH_SIZE KiB: n %d  max %.0f p90 %.0f median %.0f mean %.0f' % (len(v), v[-1], v[int(len(v)*0.9)], v[len(v)//2], sum(v)/len(v)))
PY
done
