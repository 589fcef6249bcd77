#!/bin/bash
# Round-6 eighth GPU session: pass A with the next stream-K segment's first steps requested in front of the tile stores (default
# build, CNMF_G2_PIPE=1) against the unpipelined build (tools/bin/libcnmf_nopipe.so), one box, alternating; bit identity first.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
python - <<P
import os, numpy as np, subprocess, sys
code = '''
import numpy as np, sys
from cnmf_amd import synth
from cnmf_amd.engine import Engine
X = synth.make_config("C3", dtype=np.float32, n_cells=20000)
eng = Engine(0); eng.set_matrix(X)
rs = np.random.RandomState(3)
out = {}
for tag, n, kc in (("wide", 140, 0), ("narrow", 40, 256)):
    ks = [int(k) for k in rs.randint(5, 17, size=n)]; seeds = [int(s) for s in rs.randint(1, 2**31-1, size=n)]
    H, _, it, v = eng.nmf_batch(ks, seeds=seeds, max_iter=80, warn=False, kc_max=kc)
    out[tag + "_n"] = it; out[tag + "_v"] = v; out[tag + "_H"] = np.concatenate([h.ravel() for h in H]); out[tag + "_kc"] = np.array([eng.last_stats["kc"], eng.last_stats["gemm_mode"]])
np.savez(sys.argv[1], **out)
'''
open("/tmp/pipe_run.py", "w").write(code)
root = os.environ["GRAFT_REPO_ROOT"]
for tag, lib in (("pipe", ""), ("nopipe", root + "/tools/bin/libcnmf_nopipe.so")):
    env = dict(os.environ, PYTHONPATH=root)
    if lib: env["CNMF_LIB_PATH"] = lib
    subprocess.run([sys.executable, "/tmp/pipe_run.py", "/tmp/pipe_%s.npz" % tag], check=True, env=env)
a, b = np.load("/tmp/pipe_pipe.npz"), np.load("/tmp/pipe_nopipe.npz")
print("geometry", a["wide_kc"], a["narrow_kc"], "bit identity of the pipelined pass A:", all(np.array_equal(a[k], b[k]) for k in a.files))
P
for rep in 1 2 3; do
  for lib in "" tools/bin/libcnmf_nopipe.so; do
    CNMF_LIB_PATH=${lib:+$GRAFT_REPO_ROOT/$lib} timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r6_pipe.json 2>> gpurun_out/r6_pipe.err
    python - <<P
import json
d = json.loads(open("gpurun_out/r6_pipe.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("lib=${lib:-default (pipelined)} rep $rep:", round(d["value"], 1), "restarts/s; e2e", round(r["end_to_end"]["frac"], 4), "pass A/B ms", {k: round(v, 4) for k, v in r["avg_launch_ms"].items()})
P
  done
done 2>&1 | tee gpurun_out/r6_passA_pipeline_ab.txt
