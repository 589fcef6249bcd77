#!/bin/bash
# the other BASELINE configurations through bench.py (one GPU): C3 with n_iter = 200 (1800 restarts), C4 (200 000 cells, K = 20, 100 restarts), C2 (2700 x 2000, K = 10, 100 restarts)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python bench.py --steps 1 --warmup 1 --restarts-per-k 200 --no-cpu-baseline --no-extras > gpurun_out/r3_cfg_c3_1800.json 2> gpurun_out/r3_cfg.err; echo "c3x200 rc=$?"
timeout 900 python bench.py --steps 1 --warmup 1 --workload C4 --kmin 20 --kmax 20 --no-cpu-baseline --no-extras > gpurun_out/r3_cfg_c4.json 2>> gpurun_out/r3_cfg.err; echo "c4 rc=$?"
timeout 300 python bench.py --steps 3 --warmup 1 --workload C2 --kmin 10 --kmax 10 --no-cpu-baseline --no-extras > gpurun_out/r3_cfg_c2.json 2>> gpurun_out/r3_cfg.err; echo "c2 rc=$?"
python - <<'PY'
import json
for f in ("c3_1800", "c4", "c2"):
    try:
        d = json.loads(open("gpurun_out/r3_cfg_%s.json" % f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, "restarts/s %.1f" % d["value"], "ms/step %.0f" % d["ms_per_step"], d["config"].get("workload"), "kc", d["config"].get("packed_columns"),
              "frac %.3f e2e %.3f gemm share %.3f" % (r["frac"], r["end_to_end"]["frac"], r["gemm_share_of_gpu_time"]), "util", d["config"].get("column_utilisation"))
    except Exception as e:
        print(f, "failed", e)
PY
tail -3 gpurun_out/r3_cfg.err
