#!/bin/bash
# Where the GEMM passes' wave cycles go (round 6): SQ wave-state counters of the bench's own step, separate `rocprofv3 --pmc`
# passes (kernel trace only).  MI355X_MICROARCH.md: SQ_WAIT_ANY = wave parked (s_waitcnt / barrier), SQ_WAIT_INST_ANY = issue
# stall (MFMA RAW / pipe busy), SQ_ACTIVE_INST_ANY = issuing; the three are disjoint and sum to ~SQ_WAVE_CYCLES (quad-cycles).
#   PMC_ENV="CNMF_NO_COUNTS=1" for the general path.  -> gpurun_out/${PMC_OUT:-pmc_stalls.json}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmcs; mkdir -p $R/gpurun_out/pmcs
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  env $PMC_ENV rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmcs/set$i -o pmc --output-format csv -- \
     python $R/bench.py --steps 1 --warmup 0 --restarts-per-k ${RPK:-30} --no-cpu-baseline --no-extras > $R/gpurun_out/pmcs/set$i.log 2>&1 || echo "set $i failed"
done
cd $R
python - <<PY
import csv, glob, collections, os, json
R=os.environ['GRAFT_REPO_ROOT']
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R+'/gpurun_out/pmcs/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name=r['Kernel_Name'].split('(')[0].replace('void ','')
        acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
out={"_source":"tools/gpu_pmc_stalls.sh: rocprofv3 --pmc (four separate passes, kernel trace only) around python bench.py --steps 1 --warmup 0 --restarts-per-k %s --no-cpu-baseline --no-extras %s; mean per launch over the launches of the kernel (full-width launches dominate)" % (os.environ.get('RPK','30'), os.environ.get('PMC_ENV',''))}
for key, part in (("passA","gemm2h_streamk_kernel"),("passB","gemm2h_kernel"),("sweepW","sweep_kernel<0, false, false,")):
    e={}
    for n,cs in acc.items():
        if part in n:
            for c,v in cs.items(): e.setdefault(c,[]).extend(v)
    m={c:sum(v)/len(v) for c,v in e.items()}
    m["launches"]=max((len(v) for v in e.values()), default=0)
    wc=m.get("SQ_WAVE_CYCLES")
    if wc:
        for c in ("SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_WAIT_INST_LDS","SQ_ACTIVE_INST_LDS","SQ_ACTIVE_INST_VMEM","SQ_ACTIVE_INST_MISC"):
            if c in m: m[c+"_over_WAVE_CYCLES"]=m[c]/wc
    out[key]=m
json.dump(out, open(R+'/gpurun_out/'+os.environ.get('PMC_OUT','pmc_stalls.json'),'w'), indent=1)
for k in ("passA","passB","sweepW"):
    print(k, {a:(round(b,4) if isinstance(b,float) and b<10 else round(b)) for a,b in out[k].items()})
PY
rm -rf $R/gpurun_out/pmcs/set*/
