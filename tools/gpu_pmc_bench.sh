#!/bin/bash
# HBM traffic of the two GEMM passes inside the real bench loop (separate --pmc passes, no trace domains)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmcb; mkdir -p $R/gpurun_out/pmcb
cd /tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  stag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmcb/$stag -o pmc --output-format csv -- \
     python $R/bench.py --steps 1 --warmup 0 --restarts-per-k 3 --no-cpu-baseline > $R/gpurun_out/pmcb/$stag.log 2>&1
done
python - <<PY
import csv, glob, collections, os, json
R=os.environ['GRAFT_REPO_ROOT']
out={}
for d in sorted(glob.glob(R+'/gpurun_out/pmcb/*/')):
    f=os.path.join(d,'pmc_counter_collection.csv')
    if not os.path.exists(f): continue
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        name=r['Kernel_Name'].split('(')[0]
        acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
    for name,cs in acc.items():
        for c,v in cs.items():
            out.setdefault(name,{})[c]={'mean':sum(v)/len(v),'n':len(v),'max':max(v)}
json.dump(out, open(R+'/gpurun_out/pmcb/summary.json','w'), indent=1)
for k,v in out.items():
    if 'gemm' in k or 'sweep' in k: print(k, {c:round(x['mean'],1) for c,x in v.items()})
PY
