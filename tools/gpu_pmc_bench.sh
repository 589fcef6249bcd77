#!/bin/bash
# HBM traffic + MFMA-busy of the GEMM passes inside the real bench loop: separate `rocprofv3 --pmc` passes (kernel trace
# only, no other trace domain), then profiles-style summary -> gpurun_out/pmc_traffic.json
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmcb; mkdir -p $R/gpurun_out/pmcb
cd /tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  stag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmcb/$stag -o pmc --output-format csv -- \
     python $R/bench.py --steps 1 --warmup 0 --restarts-per-k ${RPK:-100} --no-cpu-baseline --no-extras > $R/gpurun_out/pmcb/$stag.log 2>&1
done
cd $R
python - <<PY
import csv, glob, collections, os, json
R=os.environ['GRAFT_REPO_ROOT']
XPLANES=int(os.environ.get('PMC_XPLANES','1'))       # 1: the count path's integer plane; 2: a general matrix as two f16 planes (CNMF_NO_COUNTS=1)
NOTE=os.environ.get('PMC_NOTE','')
OUTNAME=os.environ.get('PMC_OUT','pmc_traffic.json')
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R+'/gpurun_out/pmcb/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name=r['Kernel_Name'].split('(')[0].replace('void ','')
        acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
def mean(name_part, c):
    v=[x for n,cs in acc.items() if name_part in n for x in cs.get(c,[])]
    return (sum(v)/len(v), len(v)) if v else (None, 0)
N_pad, G_pad = 50176, 2048
# the batch geometry of the profiled run, from its own JSON line (written by the WRITE_SIZE pass)
try:
    bl=[l for l in open(R+'/gpurun_out/pmcb/WRITE_SIZE.log') if l.startswith('{')][-1]; bj=json.loads(bl)
    KC=int(bj['config']['packed_columns']); NSPLIT=int(bj['config']['splitk_passB'])
except Exception:
    KC, NSPLIT = 1024, 8
out_geom={"packed_columns": KC, "splitk_passB": NSPLIT}
out={"_source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES (separate passes, --kernel-trace only) around \`bench.py --steps 1 --warmup 0 --restarts-per-k 100 --no-cpu-baseline --no-extras\` (the bench step itself: 900 restarts, wide batch of 1024 packed columns narrowing in the tail) on the default path (CNMF_GEMM3=4: count structure detected -> f16 two-plane kernels), tools/gpu_pmc_bench.sh; mean over all launches of the kernel. FETCH_SIZE/WRITE_SIZE in KiB as reported; hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 -- FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of 16-B/lane reads, LDS-DMA included; Infinity-Cache hits are counted). mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE/8 XCDs)."}
for key, part in (("passA","gemm2h_streamk_kernel"),("passB","gemm2h_kernel"),("sweepW","sweep_kernel<0, false, false,"),("split","split2h_finalize_kernel")):
    e={"kernel": part.rstrip(",")}
    for c in ("FETCH_SIZE","WRITE_SIZE","SQ_VALU_MFMA_BUSY_CYCLES","GRBM_GUI_ACTIVE"):
        m,n=mean(part,c); e[c]=m; e["launches"]=n or e.get("launches",0)
    if e["FETCH_SIZE"] is not None and e["WRITE_SIZE"] is not None:
        e["hbm_bytes_per_launch"]=(2*e["FETCH_SIZE"]+e["WRITE_SIZE"])*1024
    if e["SQ_VALU_MFMA_BUSY_CYCLES"] and e["GRBM_GUI_ACTIVE"]:
        e["mfma_busy_frac"]=e["SQ_VALU_MFMA_BUSY_CYCLES"]/(1024*e["GRBM_GUI_ACTIVE"]/8)
    out[key]=e
xplane = N_pad*G_pad*2*XPLANES
out["algorithmic_bytes_per_launch"]={
  "passA": xplane + KC*G_pad*4 + KC*N_pad*4,
  "passB": xplane + KC*N_pad*4 + NSPLIT*KC*G_pad*4,
  "note": "count plane of X (or X^T) once (2 B per element, f16) + the factor's two f16 planes once (4 B per element) + the product written once (pass A: one XHt plane of 51 MB -- the stream-K partial planes of cut tiles come on top; pass B: the split-K partial planes)"}
import sys; sys.path.insert(0, R)
from bench import source_hashes
out['kernel_source_sha256']=source_hashes()
out['geometry']=out_geom
if NOTE: out['_source'] = NOTE + ' -- ' + out['_source']
json.dump(out, open(R+'/gpurun_out/'+OUTNAME,'w'), indent=1)
print(json.dumps({k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk in('hbm_bytes_per_launch','mfma_busy_frac','launches')}) for k,v in out.items() if k!='_source'}, indent=1))
PY
# the raw per-dispatch counter CSVs (hundreds of MB at 5000+ launches per kernel) stay on the GPU box
rm -rf $R/gpurun_out/pmcb/FETCH_SIZE $R/gpurun_out/pmcb/WRITE_SIZE $R/gpurun_out/pmcb/SQ_VALU_MFMA_BUSY_CYCLES
