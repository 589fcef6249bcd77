#!/bin/bash
# Round-6 first GPU session: the new parity tests (Itakura-Saito tail, the C3 pipeline golden), the suites the round's host
# changes touch, then the A/B runs that decide two defaults: wide batches on small matrices (BASELINE config 2) and the
# XCD mapping of pass A on the general path.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_is_tail.py -x -q -s > gpurun_out/r6_is_tail.log 2>&1; echo "is_tail rc=$?" | tee gpurun_out/r6_first.status
grep -a "Itakura\|mirror class\|passed\|failed\|Error" gpurun_out/r6_is_tail.log | tail -20
timeout 900 python -m pytest tests/test_gpu_golden_big.py -x -q -s -k "C3_pipeline" > gpurun_out/r6_c3pipe.log 2>&1; echo "c3pipe rc=$?" | tee -a gpurun_out/r6_first.status
grep -a "C3 pipeline\|passed\|failed\|Error\|assert" gpurun_out/r6_c3pipe.log | tail -20
timeout 1500 python -m pytest tests/test_gpu_kl_tail.py tests/test_gpu_mu_sparse.py tests/test_gpu_mu.py tests/test_gpu_determinism.py tests/test_gpu_configs.py tests/test_gpu_comm.py -x -q > gpurun_out/r6_suite_a.log 2>&1; echo "suite_a rc=$?" | tee -a gpurun_out/r6_first.status
tail -5 gpurun_out/r6_suite_a.log
for ws in 0 1; do
  CNMF_WIDE_SMALL=$ws timeout 300 python bench.py --steps 5 --warmup 2 --workload C2 --kmin 10 --kmax 10 --no-cpu-baseline --no-extras > gpurun_out/r6_c2_wide$ws.json 2> gpurun_out/r6_c2_wide$ws.err
  python - <<P
import json
d = json.loads(open("gpurun_out/r6_c2_wide$ws.json").read().strip().splitlines()[-1])
print("C2 CNMF_WIDE_SMALL=$ws:", round(d["ms_per_step"], 2), "ms per job,", round(d["value"]), "restarts/s, kc", d["config"]["packed_columns"], "frac", round(d["roofline"]["frac"], 3), "e2e", round(d["roofline"]["end_to_end"]["frac"], 3))
P
done
for xm in 1 0 1 0; do
  CNMF_G2_XMAP=$xm timeout 300 python tools/general_ab.py 20 > gpurun_out/r6_general_xmap${xm}.json 2>> gpurun_out/r6_general.err
  python - <<P
import json
d = json.loads(open("gpurun_out/r6_general_xmap${xm}.json").read().strip().splitlines()[-1])
print("general CNMF_G2_XMAP=$xm:", round(d["restarts_per_s"], 1), "restarts/s, pass A/B ms", round(d["avg_launch_ms"]["passA"], 4), round(d["avg_launch_ms"]["passB"], 4), "frac", round(d["frac"], 3))
P
done
