#!/bin/bash
# kernel-trace summaries of the consensus core (C5) and of the MU solver -> gpurun_out/kernel_stats_{consensus,mu}.txt
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
rm -rf $R/gpurun_out/profc $R/gpurun_out/profm
( cd /tmp && CPU=0 CNMF_DEBUG=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/profc -o trace -- python $R/tools/gpu_cons.py > $R/gpurun_out/profc.log 2>&1 )
tail -40 $R/gpurun_out/profc.log | grep -v "^$" | tail -32
python tools/export_profile.py $(ls gpurun_out/profc/*results.db | head -1) gpurun_out/kernel_stats_consensus.txt "tools/gpu_cons.py: 3 x consensus core on C5 (5000 spectra x 2000 genes, k=20) + 1 x stats mode" | head -24
( cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/profm -o trace -- python $R/tools/gpu_mu_probe.py > $R/gpurun_out/profm.log 2>&1 )
tail -4 $R/gpurun_out/profm.log
python tools/export_profile.py $(ls gpurun_out/profm/*results.db | head -1) gpurun_out/kernel_stats_mu.txt "tools/gpu_mu_probe.py: KL multiplicative update, C3 (50000 x 2000), k = 5, 9, 13, 20, 100 iterations each" | head -12
