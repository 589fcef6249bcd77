#!/bin/bash
# KL multiplicative updates (matrix-pipe path): timing probe, kernel trace, PMC counters -> gpurun_out/
mkdir -p gpurun_out; export TMPDIR=/tmp
MU_ITERS=100 timeout 300 python tools/gpu_mu_probe.py > gpurun_out/mu_probe.log 2>&1; cat gpurun_out/mu_probe.log
CNMF_MU_VALU=1 MU_ITERS=20 timeout 300 python tools/gpu_mu_probe.py 2>&1 | head -3 | sed 's/^/[vector-ALU path] /' | tee -a gpurun_out/mu_probe.log
rm -rf gpurun_out/profm2
MU_ITERS=30 timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/profm2 -o trace -- python tools/gpu_mu_probe.py > gpurun_out/profm2.log 2>&1
python tools/export_profile.py $(find gpurun_out/profm2 -name "*.db" | head -1) gpurun_out/mu_stats.txt "tools/gpu_mu_probe.py, MU_ITERS=30: KL multiplicative update, C3 (50000 x 2000), k = 9 / 13 / 20 alone, 16 x k=9, 36 x k=5..13, 16 x k=20 (matrix-pipe path, kernels_mu_mfma.hip.h)" | head -14
python - <<'PY'
import sqlite3, glob
c = sqlite3.connect(glob.glob("gpurun_out/profm2/**/*.db", recursive=True)[0])
rows = list(c.execute("select name, grid_x, grid_y, grid_z, count(*), avg(end-start)/1e3 from kernels where name like '%coop_kernel%' group by name, grid_x, grid_y, grid_z order by name, grid_z"))
with open("gpurun_out/mu_stats.txt", "a") as f:
    f.write("\n# per launch shape (grid in threads; grid_z = restart groups of 4): average duration\n")
    for r in rows:
        f.write("%-60s grid (%d,%d,%d) n=%d avg %.1f us\n" % (r[0].split("(")[0][-60:], r[1], r[2], r[3], r[4], r[5]))
print(open("gpurun_out/mu_stats.txt").read()[-900:])
PY
bash tools/gpu_r2_mu_pmc.sh 2>&1 | tail -4 | tee gpurun_out/mu_pmc.txt
