#!/bin/bash
# Round-5 second GPU session: KL tail + non-zero images from the compressed rows, the store / ABI suites, C4 iteration counts,
# the GEMM ablation at the production launch shape beside the bare MFMA stream, and the default bench line.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kl_tail.py tests/test_gpu_mu_sparse.py tests/test_gpu_comm.py tests/test_abi.py "tests/test_gpu_pipeline.py::test_merged_spectra_served_from_the_device_store" -x -q -s > gpurun_out/r5_second_tests.log 2>&1; echo "tests rc=$?" | tee gpurun_out/r5_second.status
grep -v "^$" gpurun_out/r5_second_tests.log | grep -i "sum of squared\|mirror class\|passed\|failed\|error\|worst" | tail -20
WORKLOAD=C4 KMIN=20 KMAX=20 N_ITER=100 timeout 600 python tools/dump_iters.py > gpurun_out/r5_iters_c4.log 2>&1; echo "iters rc=$?" | tee -a gpurun_out/r5_second.status
tail -3 gpurun_out/r5_iters_c4.log
timeout 300 python tools/probe_gemm2h_ablate.py > gpurun_out/r5_gemm2h_ablation.txt 2>&1; echo "ablate rc=$?" | tee -a gpurun_out/r5_second.status
cat gpurun_out/r5_gemm2h_ablation.txt | tail -24
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/mfma_stream_probe.hip -o /tmp/mfma_stream_probe && /tmp/mfma_stream_probe > gpurun_out/r5_mfma_stream.txt 2>&1; cat gpurun_out/r5_mfma_stream.txt | tail -8
timeout 900 python bench.py > gpurun_out/r5_bench_default.json 2> gpurun_out/r5_bench_default.err; echo "bench rc=$?" | tee -a gpurun_out/r5_second.status
python - <<P
import json
d = json.loads(open("gpurun_out/r5_bench_default.json").read().strip().splitlines()[-1])
print("bench:", round(d["value"], 1), "restarts/s; roofline", round(d["roofline"]["frac"], 3), "passA/B TF", round(d["roofline"]["achieved_passA"]), round(d["roofline"]["achieved_passB"]), "e2e", round(d["roofline"]["end_to_end"]["frac"], 3), "gemm share", round(d["roofline"]["gemm_share_of_gpu_time"], 3), "tail", round(d["config"]["tail"]["share_of_gpu_time"], 3))
print("hints:", d.get("with_queue_hints"))
print("kl:", json.dumps(d.get("kl_non_zero_path"))[:900])
print("e2e:", d["e2e"]["stages_s"], d["e2e"]["total_s"])
P
