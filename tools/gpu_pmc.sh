#!/bin/bash
# PMC passes for the two GEMM kernels (counters only: no sys/hip trace domains with --pmc)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
cd /tmp
rocprofv3 -L 2>/dev/null | grep -E "^\s*(Name|gpu-agent0)|SQ_VALU_MFMA|GRBM_GUI|SQ_BUSY_CY|SQ_WAVE_CYCLES|SQ_WAIT|SQ_ACTIVE_INST|SQ_INSTS_VALU_MFMA|SQ_LDS|FETCH_SIZE|WRITE_SIZE|MfmaUtil|TCC_HIT|TCC_MISS|TCP_" | head -80 > $R/gpurun_out/pmc/counters_list.txt
for cfg in "0 1 256 1" "0 2 256 1" "1 2 256 16"; do
  tag=$(echo $cfg | tr ' ' '_')
  for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
    stag=$(echo $set | cut -d' ' -f1)
    REPS=3 rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmc/${tag}_${stag} -o pmc --output-format csv -- python $R/tools/prof_gemm.py $cfg > $R/gpurun_out/pmc/${tag}_${stag}.log 2>&1
  done
done
ls -R $R/gpurun_out/pmc | head -50
