"""Race / determinism soak of the round-3 kernels: multiplicative updates at padded rank 64 (two restarts per workgroup),
the one-workgroup k-means++ kernels (registers: 1 / 2 / 4 / 8 values per thread; global memory), the register KNN selection,
the triangular distance kernel, the device-side Lloyd stopping rule -- repeated calls must return bit-identical results."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cnmf_amd import synth
from cnmf_amd.engine import Engine
reps = int(os.environ.get("REPS", "6"))
eng = Engine(0)
X = synth.make_config("C3", dtype=np.float32, n_cells=12000)
eng.set_matrix(X)
ks = [40, 33, 64, 48, 57, 36, 9, 20, 64]
seeds = list(range(300, 300 + len(ks)))
ref = None
for rep in range(reps):
    H, W, n, err = eng.nmf_mu_batch(ks, seeds=seeds, max_iter=30, return_W=True, warn=False)
    Hi, Wi, ni, erri = eng.nmf_mu_batch([40, 64, 50], seeds=[1, 2, 3], beta_loss="itakura-saito", max_iter=20, return_W=True, warn=False)
    cur = [np.concatenate([h.ravel() for h in H]), np.concatenate([w.ravel() for w in W]), n.copy(), np.asarray(err),
           np.concatenate([h.ravel() for h in Hi]), np.concatenate([w.ravel() for w in Wi])]
    if ref is None:
        ref = cur
    else:
        same = [np.array_equal(a, b) for a, b in zip(ref, cur)]
        print("MU rep %d identical: %s" % (rep, same), flush=True)
        assert all(same)
cases = [(700, 64, 5, 20), (1800, 96, 7, 40), (3500, 128, 9, 60), (5000, 256, 20, 100), (9000, 48, 6, 150)]
for (R, G, k, nout) in cases:
    S, _ = synth.consensus_stress(R=R, G=G, k=k, n_outliers=nout, seed=R)
    ref = None
    for rep in range(reps):
        out = eng.consensus(S, k, density_threshold=0.5)
        st = eng.consensus(S, k, skip_density=True, want_silhouette=True)
        cur = [out["local_density"], out["labels"], out["median_spectra"], np.float64(out["inertia"]), st["labels"], np.float64(st["silhouette"])]
        if ref is None:
            ref = cur
        else:
            same = [np.array_equal(a, b) for a, b in zip(ref, cur)]
            assert all(same), (R, rep, same)
    print("consensus R=%d k=%d: %d repetitions identical" % (R, k, reps), flush=True)
print("round-3 determinism soak ok")
