#!/bin/bash
# Round-6 sixth GPU session: the W half-step with the next chunk's loads requested ahead (tools/bin/libcnmf_swpf.so,
# -DCNMF_SWEEP_PF=1) against the default build, one box, alternating; bit identity first.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
python - <<P
import os, numpy as np, subprocess, sys, json
code = '''
import numpy as np, sys
from cnmf_amd import synth
from cnmf_amd.engine import Engine
X = synth.make_config("C3", dtype=np.float32, n_cells=20000)
eng = Engine(0); eng.set_matrix(X)
rs = np.random.RandomState(3)
ks = [int(k) for k in rs.randint(5, 17, size=140)]; seeds = [int(s) for s in rs.randint(1, 2**31-1, size=140)]
H, _, n, v = eng.nmf_batch(ks, seeds=seeds, max_iter=80, warn=False)
np.savez(sys.argv[1], n=n, v=v, H=np.concatenate([h.ravel() for h in H]))
'''
open("/tmp/pf_run.py", "w").write(code)
root = os.environ["GRAFT_REPO_ROOT"]
for tag, lib in (("base", ""), ("pf", root + "/tools/bin/libcnmf_swpf.so")):
    env = dict(os.environ, PYTHONPATH=root)
    if lib: env["CNMF_LIB_PATH"] = lib
    subprocess.run([sys.executable, "/tmp/pf_run.py", "/tmp/pf_%s.npz" % tag], check=True, env=env)
a, b = np.load("/tmp/pf_base.npz"), np.load("/tmp/pf_pf.npz")
print("bit identity of the prefetching sweep:", bool(np.array_equal(a["n"], b["n"]) and np.array_equal(a["v"], b["v"]) and np.array_equal(a["H"], b["H"])))
P
for rep in 1 2; do
  for lib in "" tools/bin/libcnmf_swpf.so; do
    CNMF_LIB_PATH=${lib:+$GRAFT_REPO_ROOT/$lib} timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r6_swpf.json 2>> gpurun_out/r6_swpf.err
    python - <<P
import json
d = json.loads(open("gpurun_out/r6_swpf.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("lib=${lib:-default} rep $rep:", round(d["value"], 1), "restarts/s; e2e", round(r["end_to_end"]["frac"], 4), "gemm share", round(r["gemm_share_of_gpu_time"], 4), "pass A/B ms", {k: round(v, 4) for k, v in r["avg_launch_ms"].items()})
P
  done
done 2>&1 | tee gpurun_out/r6_sweep_prefetch_ab.txt
