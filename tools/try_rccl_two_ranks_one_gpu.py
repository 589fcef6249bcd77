"""Can the in-library RCCL communicator run with TWO ranks on ONE GPU (the only multi-rank configuration a 1-GPU box
offers)?  Spawns two processes, both on device 0, through the torch-free file rendezvous, and reports what RCCL says."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
from cnmf_amd.engine import Engine
from cnmf_amd import dist as cd
rank, world, path = int(sys.argv[1]), 2, sys.argv[2]
eng = Engine(0)
try:
    cd.comm_bootstrap_file(eng, rank, world, path, timeout=60)
    v = eng.allgather_array(np.array([rank + 1.0]))
    print("rank", rank, "allgather ->", v.ravel().tolist(), flush=True)
except Exception as e:
    print("rank", rank, "FAILED:", type(e).__name__, str(e)[:300], flush=True)
''' % ROOT
d = tempfile.mkdtemp()
env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="WARN")
ps = [subprocess.Popen([sys.executable, "-c", CHILD, str(r), os.path.join(d, "id")], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True) for r in range(2)]
for p in ps:
    try:
        out, _ = p.communicate(timeout=120)
    except subprocess.TimeoutExpired:
        p.kill(); out = "(timeout)"
    print(out[-1200:])
