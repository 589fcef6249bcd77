#!/bin/bash
# Round-5 fifth GPU session: the full-size KL golden test + the new edge tests, cheap A/Bs of two host knobs on the final
# kernels (snapshot lag, sweep workgroups per slot), PMC traffic of the general path.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_kl_tail.py tests/test_gpu_comm.py tests/test_gpu_golden_big.py -k "kl or comm or C4_kl" -x -q -s > gpurun_out/r5_fifth_tests.log 2>&1; echo "tests rc=$?" | tee gpurun_out/r5_fifth.status
grep -v "^$" gpurun_out/r5_fifth_tests.log | grep -i "C4 KL\|passed\|failed\|error\|assert" | tail -20
for cfg in "base" "CNMF_LAG=1" "CNMF_LAG=3" "CNMF_SWEEP_PARTS=49" "CNMF_SWEEP_PARTS=98" "base"; do
  if [ "$cfg" = "base" ]; then env python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r5_ab.json 2> gpurun_out/r5_ab.err
  else env $cfg python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r5_ab.json 2> gpurun_out/r5_ab.err; fi
  python - "$cfg" <<'P'
import json, sys
try:
    d = json.loads(open("gpurun_out/r5_ab.json").read().strip().splitlines()[-1])
    print("A/B %-22s %.1f restarts/s  e2e frac %.3f  gemm share %.3f  util %.4f" % (sys.argv[1], d["value"], d["roofline"]["end_to_end"]["frac"], d["roofline"]["gemm_share_of_gpu_time"], d["config"]["column_utilisation"]))
except Exception as e:
    print("A/B", sys.argv[1], "failed", e)
P
done | tee gpurun_out/r5_knob_ab.txt
CNMF_NO_COUNTS=1 PMC_XPLANES=2 PMC_NOTE="general path (CNMF_NO_COUNTS=1: X as two f16 planes, gemm_mode 5)" RPK=30 PMC_OUT=r5_pmc_traffic_general.json bash tools/gpu_pmc_bench.sh > gpurun_out/r5_pmc_general.log 2>&1
tail -30 gpurun_out/r5_pmc_general.log | head -24
