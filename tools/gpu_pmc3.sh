#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc3; mkdir -p $R/gpurun_out/pmc3
cd /tmp
export CNMF_GEMM3=3
for dbg in 0 6 1; do
  for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    stag=$(echo $set | cut -d' ' -f1)
    CNMF_G3_DBG=$dbg REPS=3 rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmc3/d${dbg}_${stag} -o pmc --output-format csv -- python $R/tools/prof_gemm3.py B > $R/gpurun_out/pmc3/d${dbg}_${stag}.log 2>&1
  done
done
python - <<PY
import csv, glob, collections, os
R=os.environ['GRAFT_REPO_ROOT']
for d in sorted(glob.glob(R+'/gpurun_out/pmc3/*/')):
    dur={}
    for f in glob.glob(d+'/**/pmc_kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'gemm3' in r['Kernel_Name']:
                dur.setdefault('dur_us',[]).append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
    for f in glob.glob(d+'/**/pmc_counter_collection.csv', recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            name=r['Kernel_Name'].split('(')[0]
            if 'gemm3' not in name: continue
            acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
        for name,cs in acc.items():
            print(os.path.basename(d.rstrip('/')), name, {c:round(sum(v)/len(v),1) for c,v in cs.items()}, 'dur_us', [round(x,1) for x in dur.get('dur_us',[])])
PY
