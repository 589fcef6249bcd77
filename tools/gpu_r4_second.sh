#!/bin/bash
# round 4, second GPU session: the new parity tests (f64 consensus tail, NNDSVD on rank-deficient blocks, drift-calibrated
# stopping rule) and the batch-width experiment (1024 / 1536 / 2048 packed columns)
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_tail.py "tests/test_gpu_nmf.py" -k "tail or nndsvd or nnls or golden_reference or mirror or usage" -x -q -s 2>&1 | tail -40 > gpurun_out/r4_tests_a.log
tail -25 gpurun_out/r4_tests_a.log
timeout 900 python -m pytest tests/test_gpu_golden_big.py::test_C3_long_restarts_vs_sklearn_golden -x -q -s 2>&1 | tail -30 > gpurun_out/r4_tests_b.log
tail -20 gpurun_out/r4_tests_b.log
for lim in 1024 1536 2048; do
  CNMF_KC_LIMIT=$lim timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r4_bench_kc$lim.json 2> gpurun_out/r4_bench_kc$lim.err
  python - <<P
import json
try:
    d = json.loads(open("gpurun_out/r4_bench_kc$lim.json").read().strip().splitlines()[-1])
    print("KC limit $lim:", d["value"], "restarts/s", d["ms_per_step"], "ms", {k: d["config"].get(k) for k in ("packed_columns", "column_utilisation")}, d["roofline"]["gemm_share_of_gpu_time"])
except Exception as e:
    print("KC limit $lim failed", e); print(open("gpurun_out/r4_bench_kc$lim.err").read()[-1500:])
P
done
