"""Race / determinism soak of the round-5 paths: a CSR upload whose compressed rows, device-built transpose and non-zero
images are REBUILT in every repetition (a fresh upload each time) must give bit-identical Kullback-Leibler restarts
(non-zero kernels), float64 refits (both orientations, column divisor), gene statistics and z-scored OLS products --
the counting-sort transpose (csr_host.hip.h) relies on a fixed order, the refits on fixed-order butterfly sums."""
import os, sys
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cnmf_amd import synth
from cnmf_amd.engine import Engine
C, _ = synth.topic_counts(int(os.environ.get("N_CELLS", "60000")), 2000, 12, 5.0, 0.4, 9)
X = sp.csr_matrix(synth.normalise_like_prepare(C, dtype=np.float32))
print("matrix", X.shape, "%.1f %% non-zero" % (100.0 * X.nnz / np.prod(X.shape)), flush=True)
eng = Engine(0)
rs = np.random.RandomState(0)
ks = [5, 9, 13, 7, 20, 16, 3, 24, 11, 32]
H9 = np.abs(rs.standard_normal((9, X.shape[1])))
U9 = np.abs(rs.standard_normal((9, X.shape[0])))
Wd = np.abs(rs.standard_normal((X.shape[0], 9)))
div = np.zeros(X.shape[1]); sel = np.sort(rs.choice(X.shape[1], 700, replace=False)); div[sel] = rs.uniform(0.5, 2.0, 700)
ref = None
for rep in range(int(os.environ.get("REPS", "4"))):
    eng.set_matrix(X)
    H, W, n, err = eng.nmf_mu_batch(ks, seeds=list(range(100, 100 + len(ks))), max_iter=40, return_W=True, warn=False)
    assert not eng.matrix_images()["dense"]
    Wr, nr, er = eng.mu_refit_f64(H9, max_iter=50, warn=False)
    Wt, nt, et = eng.mu_refit_f64(U9, transposed=True, max_iter=50, warn=False)
    Wc, nc, ec = eng.mu_refit_f64(H9 * (div != 0), col_divisor=div, w_init=0.3, n_features=700, max_iter=30, warn=False)
    mean, var = eng.col_mean_var()
    xty = eng.xt_matmul_f64(Wd, mean=mean, std=np.sqrt(np.where(var < 1e-12, 1e-12, var)))
    cur = (np.concatenate([h.ravel() for h in H]), np.concatenate([w.ravel() for w in W]), n.copy(), np.asarray(err),
           Wr, Wt, Wc, np.array([nr, nt, nc]), np.array([er, et, ec]), mean, var, xty)
    if ref is None:
        ref = cur
    else:
        same = [np.array_equal(a, b) for a, b in zip(ref, cur)]
        print("rep %d identical: %s" % (rep, same), flush=True)
        assert all(same)
print("round-5 determinism soak ok")
