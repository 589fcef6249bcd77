"""Race / determinism soak: the same batched calls repeated must return bit-identical results (a missing barrier or an
LDS double-buffer hazard in the cooperative kernels would show up as run-to-run differences)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cnmf_amd import synth
from cnmf_amd.engine import Engine
X = synth.make_config("C3", dtype=np.float32, n_cells=int(os.environ.get("N_CELLS", "20000")))
eng = Engine(0); eng.set_matrix(X)
ks = [5, 9, 13, 7, 11, 6, 12, 8, 10, 5, 9, 13, 20, 17, 24, 32, 3, 16, 1]
seeds = list(range(100, 100 + len(ks)))
ref = None
for rep in range(int(os.environ.get("REPS", "4"))):
    H, W, n, err = eng.nmf_mu_batch(ks, seeds=seeds, max_iter=40, return_W=True, warn=False)
    Hi, Wi, ni, erri = eng.nmf_mu_batch(ks[:7], seeds=seeds[:7], beta_loss="itakura-saito", max_iter=20, return_W=True, warn=False)
    Hc, _, nc, viol = eng.nmf_batch(ks * 6, seeds=list(range(500, 500 + 6 * len(ks))), max_iter=60, warn=False)
    cur = (np.concatenate([h.ravel() for h in H]), np.concatenate([w.ravel() for w in W]), n.copy(), np.asarray(err),
           np.concatenate([h.ravel() for h in Hi]), np.concatenate([w.ravel() for w in Wi]),
           np.concatenate([h.ravel() for h in Hc]), nc.copy())
    if ref is None:
        ref = cur
    else:
        same = [np.array_equal(a, b) for a, b in zip(ref, cur)]
        print("rep %d identical: %s" % (rep, same), flush=True)
        assert all(same)
print("determinism soak ok")
