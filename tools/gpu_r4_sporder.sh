#!/bin/bash
# conflict-free entry order of the KL non-zero path: parity tests, A/B against the storage order, LDS counters
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_mu_sparse.py -x -q > gpurun_out/r4_sporder_tests.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r4_sporder_tests.log
for o in 0 1 0 1; do
  echo "CNMF_SP_ORDER=$o"
  CNMF_SP_ORDER=$o SP_ONLY=1 SP_LONG=1 SP_MODES=1 MU_ITERS=150 timeout 600 python tools/mu_sparse_probe.py 2>&1 | grep "us per\|first call"
done > gpurun_out/r4_mu_sparse_order_ab.txt 2>&1
cat gpurun_out/r4_mu_sparse_order_ab.txt
rm -rf /tmp/sppmc; cd /tmp
SP_ONLY=1 SP_LONG=1 MU_ITERS=10 timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d /tmp/sppmc -o pmc --output-format csv -- python $R/tools/mu_sparse_probe.py > $R/gpurun_out/r4_mu_sparse_pmc2.log 2>&1; echo "pmc rc=$?"
cd $R
python - <<'PY'
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/sppmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if 'mu_sp_kernel' in name or 'sp_reorder' in name:
            acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
out = {"_source": "as r4_mu_sparse_pmc.json, with the conflict-free entry order (sp_reorder_kernel)"}
for name, cs in sorted(acc.items()):
    e = {c: sum(v) / len(v) for c, v in cs.items()}
    e["launches"] = max(len(v) for v in cs.values())
    cyc = e.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if e.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_conflict_share_of_lds_cycles"] = e.get("SQ_LDS_BANK_CONFLICT", 0.0) / e["SQ_LDS_IDX_ACTIVE"]
    if cyc:
        e["lds_busy_frac"] = e.get("SQ_LDS_IDX_ACTIVE", 0.0) / (256.0 * cyc)
        e["valu_issue_frac"] = e.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / (1024.0 * cyc)
    out[name] = e
json.dump(out, open('gpurun_out/r4_mu_sparse_pmc_ordered.json', 'w'), indent=1)
for k, v in out.items():
    if k != "_source": print(k, {a: round(b, 3) for a, b in v.items() if a in ("lds_conflict_share_of_lds_cycles", "lds_busy_frac", "valu_issue_frac", "launches")}, "cycles %.3g" % (v.get("GRBM_GUI_ACTIVE", 0) / 8))
PY
