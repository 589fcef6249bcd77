#!/bin/bash
# last session of round 4: counters + steady-state timing of the KL non-zero path, then the whole GPU suite, smoke and
# the default bench line on the final build
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
# 1. LDS / VALU counters of the non-zero kernels (kernel trace only)
rm -rf /tmp/sppmc; cd /tmp
SP_ONLY=1 SP_LONG=1 MU_ITERS=10 timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d /tmp/sppmc -o pmc --output-format csv -- python $R/tools/mu_sparse_probe.py > $R/gpurun_out/r4_mu_sparse_pmc.log 2>&1; echo "pmc rc=$?"
cd $R
python - <<'PY'
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/sppmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if 'mu_sp_kernel' in name:
            acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
out = {"_source": "rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace around SP_ONLY=1 SP_LONG=1 MU_ITERS=10 tools/mu_sparse_probe.py (200 000 x 2 000, 9 % non-zero; 32 restarts of rank 9, 36 of ranks 5..13, 32 of rank 20); mean per launch"}
for name, cs in sorted(acc.items()):
    e = {c: sum(v) / len(v) for c, v in cs.items()}
    e["launches"] = max(len(v) for v in cs.values())
    if e.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_conflict_share_of_lds_cycles"] = e.get("SQ_LDS_BANK_CONFLICT", 0.0) / e["SQ_LDS_IDX_ACTIVE"]
    if e.get("GRBM_GUI_ACTIVE"):
        # LDS-array cycles per CU against the kernel's duration in shader cycles (256 CUs)
        e["lds_busy_frac"] = e.get("SQ_LDS_IDX_ACTIVE", 0.0) / (256.0 * e["GRBM_GUI_ACTIVE"] / 8.0) if False else None
    out[name] = e
json.dump(out, open('gpurun_out/r4_mu_sparse_pmc.json', 'w'), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
# 2. steady state: 150 iterations per restart, both paths
SP_ONLY=1 SP_LONG=1 SP_MODES=1,0 MU_ITERS=150 timeout 600 python tools/mu_sparse_probe.py > gpurun_out/r4_mu_sparse_long.txt 2>&1; echo "long rc=$?"
grep "non-zero\|us per" gpurun_out/r4_mu_sparse_long.txt
# 3. the final build: whole GPU suite, smoke, default bench line
bash tools/gpu_r4_last.sh
