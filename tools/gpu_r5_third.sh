#!/bin/bash
# Round-5 third GPU session: KL pipeline, the C4 tests (count-valued 10-iteration golden + the new stopping-rule golden at the
# default width), the other BASELINE configurations through bench.py (C4 K = 20, C2 K = 10, C3 with n_iter = 200).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest "tests/test_gpu_kl_tail.py::test_kl_pipeline_factorize_to_consensus_vs_reference" tests/test_gpu_golden_big.py -k "kl_pipeline or C4_count or C4_as" -x -q -s > gpurun_out/r5_third_tests.log 2>&1; echo "tests rc=$?" | tee gpurun_out/r5_third.status
grep -v "^$" gpurun_out/r5_third_tests.log | grep -i "C4 K=20\|Kullback\|passed\|failed\|error\|assert" | tail -20
timeout 900 python bench.py --steps 2 --warmup 1 --workload C4 --kmin 20 --kmax 20 --no-cpu-baseline --no-extras > gpurun_out/r5_cfg_c4.json 2> gpurun_out/r5_cfg.err; echo "c4 rc=$?" | tee -a gpurun_out/r5_third.status
timeout 300 python bench.py --steps 3 --warmup 1 --workload C2 --kmin 10 --kmax 10 --no-cpu-baseline --no-extras > gpurun_out/r5_cfg_c2.json 2>> gpurun_out/r5_cfg.err; echo "c2 rc=$?" | tee -a gpurun_out/r5_third.status
timeout 600 python bench.py --steps 1 --warmup 1 --restarts-per-k 200 --no-cpu-baseline --no-extras > gpurun_out/r5_cfg_c3_1800.json 2>> gpurun_out/r5_cfg.err; echo "c3x200 rc=$?" | tee -a gpurun_out/r5_third.status
python - <<'PY'
import json
for f in ("c3_1800", "c4", "c2"):
    try:
        d = json.loads(open("gpurun_out/r5_cfg_%s.json" % f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, "restarts/s %.1f" % d["value"], "ms/step %.0f" % d["ms_per_step"], "kc", d["config"].get("packed_columns"),
              "frac %.3f e2e %.3f gemm share %.3f" % (r["frac"], r["end_to_end"]["frac"], r["gemm_share_of_gpu_time"]), "util", d["config"].get("column_utilisation"), "mean its", d["config"].get("mean_iterations_per_restart"))
    except Exception as e:
        print(f, "failed", e)
PY
tail -3 gpurun_out/r5_cfg.err
