#!/bin/bash
# round 4, first GPU session: (1) MFMA-only streams f16 vs i8 (go/no-go input for an int8 count path),
# (2) two factorize pipelines on one GPU at the round-3 kernels, with the GEMMs on all / 224 / 208 CUs
mkdir -p gpurun_out; export TMPDIR=/tmp
./tools/bin/mfma_stream_probe 20000 256 > gpurun_out/r4_mfma_stream.json 2> gpurun_out/r4_mfma_stream.err
cat gpurun_out/r4_mfma_stream.json
for slots in 256 224 208; do
  echo "== CNMF_WG_SLOTS=$slots"
  CNMF_WG_SLOTS=$slots REPS=2 PER_K=100 timeout 400 python tools/probe_two_engines.py 2>&1 | grep engine | tee -a gpurun_out/r4_two_engines_$slots.log
done
