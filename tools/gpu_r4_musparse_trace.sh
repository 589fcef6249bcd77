#!/bin/bash
# kernel trace of the Kullback-Leibler restarts on the non-zeros (200 000 x 2 000, 9 % non-zero)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf /tmp/sptrace
cd /tmp && SP_ONLY=1 timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/sptrace -o sp -- python $GRAFT_REPO_ROOT/tools/mu_sparse_probe.py > $GRAFT_REPO_ROOT/gpurun_out/r4_mu_sparse_trace.log 2>&1; echo "trace rc=$?"
cd $GRAFT_REPO_ROOT
grep "us per restart" gpurun_out/r4_mu_sparse_trace.log
DB=$(ls /tmp/sptrace/*/*results.db /tmp/sptrace/*results.db 2>/dev/null | head -1)
python tools/export_profile.py $DB gpurun_out/r4_kernel_stats_mu_sparse.txt "SP_ONLY=1 python tools/mu_sparse_probe.py (KL on the non-zeros, 200 000 x 2 000 at 9 % non-zero; k = 9 x 1 / 8 / 32, k = 5..13 x 36, k = 20 x 32; 30 iterations each)" > /dev/null 2>&1
head -18 gpurun_out/r4_kernel_stats_mu_sparse.txt | cut -c1-150
