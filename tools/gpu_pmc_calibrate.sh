#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of streams with KNOWN byte counts, per access width -> gpurun_out/pmc_calibration.json
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pmccal; mkdir -p $R/gpurun_out
cd /tmp
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --pmc $set --kernel-trace -d /tmp/pmccal/$set -o pmc --output-format csv -- python $R/tools/pmc_calibrate.py > /tmp/pmccal_$set.log 2>&1
done
cd $R
python - <<PY
import csv, glob, collections, json
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pmccal/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'].split('(')[0].replace('void ','')][r['Counter_Name']].append(float(r['Counter_Value']))
n_bytes = (128 << 20) * 4
out={"_source": "tools/gpu_pmc_calibrate.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) around cnmf_debug_stream: 10 launches each of a copy of 512 MiB with 4 B per lane, with 16 B per lane, and a read-only LDS-DMA stream (global_load_lds_dwordx4) of 512 MiB; counters in KiB as reported; ratio = reported bytes / true bytes",
     "true_bytes_read_per_launch": n_bytes}
for key, part, writes in (("copy_4B_per_lane","calib_copy1_kernel",True),("copy_16B_per_lane","calib_copy4_kernel",True),("ldsdma_16B_per_lane","calib_ldsdma_kernel",False)):
    e={}
    for n,cs in acc.items():
        if part in n:
            for c,v in cs.items(): e[c]=sum(v)/len(v)
    if 'FETCH_SIZE' in e: e['fetch_ratio']=e['FETCH_SIZE']*1024/n_bytes
    if writes and 'WRITE_SIZE' in e: e['write_ratio']=e['WRITE_SIZE']*1024/n_bytes
    out[key]=e
json.dump(out, open('gpurun_out/pmc_calibration.json','w'), indent=1)
print(json.dumps(out, indent=1))
PY
