#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmcm; mkdir -p $R/gpurun_out/pmcm
cat > /tmp/mu_one.py <<'PY'
import os, sys, numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from cnmf_amd import synth
from cnmf_amd.engine import Engine
X = synth.make_config("C3", dtype=np.float32)
eng = Engine(0); eng.set_matrix(X)
eng.nmf_mu_batch([9] * 16, seeds=list(range(16)), max_iter=6, tol=0, warn=False)
PY
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_ACTIVE_INST_MISC"; do
  stag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmcm/$stag -o pmc --output-format csv -- python /tmp/mu_one.py > $R/gpurun_out/pmcm/$stag.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections, os
R=os.environ['GRAFT_REPO_ROOT']
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R+'/gpurun_out/pmcm/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name=r['Kernel_Name'].split('(')[0].replace('void ','')
        if 'mu_' in name and ('coop' in name or 'mfma' in name) and 'finish' not in name: acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
for n,cs in acc.items():
    print(n[:50], {c: round(sum(v)/len(v)) for c,v in cs.items()})
PY
