#!/bin/bash
# Round-6 final check on the committed tree: the whole GPU suite, smoke, the default bench line (kernel sources unchanged since
# the closing session: its hash-stamped profiles stay valid).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r6_pytest_final.log 2>&1; echo "pytest rc=$?"
grep -a "passed\|failed" gpurun_out/r6_pytest_final.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6_smoke.log 2>&1; echo "smoke rc=$?"
bash tools/gpu_r6_bench_line.sh
python - <<P
import json
d = json.loads(open("gpurun_out/r6_bench_default.json").read().strip().splitlines()[-1])
print("e2e:", {k: round(v, 3) for k, v in d["e2e"]["stages_s"].items()}, round(d["e2e"]["total_s"], 3), d["e2e"].get("consensus_spectra_sumsq_vs_cpu"))
P
