#!/bin/bash
# Round-6 second GPU session: the failed tests of the first one re-run, the general path's spread instruction stream
# (bit identity against the burst loop, then the A/B on one box), the whole mode-5 test set.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_is_tail.py tests/test_gpu_kl_tail.py -x -q > gpurun_out/r6_tails.log 2>&1; echo "tails rc=$?" | tee gpurun_out/r6_second.status
tail -4 gpurun_out/r6_tails.log
timeout 900 python -m pytest tests/test_gpu_nmf.py -x -q -k "general or gemm_mode or wide_batch" > gpurun_out/r6_general_tests.log 2>&1; echo "general tests rc=$?" | tee -a gpurun_out/r6_second.status
tail -6 gpurun_out/r6_general_tests.log
timeout 900 python -m pytest tests/test_gpu_golden_big.py -x -q -k "C4_csr" > gpurun_out/r6_c4csr.log 2>&1; echo "c4csr rc=$?" | tee -a gpurun_out/r6_second.status
tail -3 gpurun_out/r6_c4csr.log
for gv in 4 0 4 0; do
  CNMF_G2_GVAR=$gv timeout 300 python tools/general_ab.py 20 > gpurun_out/r6_general_gvar${gv}.json 2>> gpurun_out/r6_general.err
  python - <<P
import json
d = json.loads(open("gpurun_out/r6_general_gvar${gv}.json").read().strip().splitlines()[-1])
print("general CNMF_G2_GVAR=$gv:", round(d["restarts_per_s"], 1), "restarts/s, pass A/B ms", round(d["avg_launch_ms"]["passA"], 4), round(d["avg_launch_ms"]["passB"], 4), "frac", round(d["frac"], 3))
P
done
