#!/bin/bash
# pipeline tests + the default bench line (e2e leg with the factorize host seconds)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_comm.py tests/test_gpu_tail.py -m gpu -x -q > gpurun_out/r3_e2e_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3_e2e_pytest.log
timeout 900 python bench.py > gpurun_out/r3_bench_default.json 2> gpurun_out/r3_bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['e2e']['stages_s'], d['e2e']['total_s'], d['e2e'].get('factorize_host_s'))
PY
