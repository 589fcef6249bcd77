"""Round 5: what the float64 Kullback-Leibler refits cost at BASELINE config 4's shape (200 000 x 2000, ~9 % non-zero, CSR
upload): compressed rows of X^T built on the device, refit_usage (rows = cells) and refit_spectra (rows = genes) at
K = 9 and K = 20, 100 iterations each (tol = 0), the gene statistics and the z-scored OLS product on the stored entries."""
import os, sys, time
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cnmf_amd import synth
from cnmf_amd.engine import Engine

C, _ = synth.topic_counts(int(os.environ.get("N_CELLS", "200000")), 2000, 20, 5.2, 0.4, 3)
X = sp.csr_matrix(synth.normalise_like_prepare(C, dtype=np.float32))
del C
print("matrix %d x %d, %d stored entries (%.1f %%)" % (X.shape[0], X.shape[1], X.nnz, 100.0 * X.nnz / np.prod(X.shape)), flush=True)
eng = Engine(0)
t = time.perf_counter(); eng.set_matrix(X); print("CSR upload %.3f s" % (time.perf_counter() - t), flush=True)
rs = np.random.RandomState(0)
for k in (9, 20):
    H = np.abs(rs.standard_normal((k, X.shape[1])))
    U = np.abs(rs.standard_normal((k, X.shape[0])))
    for label, arg, kw in (("refit_usage   (rows = cells)", H, {}), ("refit_spectra (rows = genes)", U, dict(transposed=True))):
        eng.mu_refit_f64(arg, max_iter=1, tol=0.0, warn=False, **kw)           # (first call of an orientation builds its rows)
        t = time.perf_counter()
        W, n, err = eng.mu_refit_f64(arg, max_iter=100, tol=0.0, warn=False, **kw)
        dt = time.perf_counter() - t
        print("k=%2d %s: %d iterations in %.3f s = %.2f ms per iteration (%.1f G stored entries/s), err %.6g"
              % (k, label, n, dt, 1e3 * dt / n, X.nnz * n / dt / 1e9, err), flush=True)
t = time.perf_counter(); mean, var = eng.col_mean_var(); t1 = time.perf_counter() - t
Wd = np.abs(rs.standard_normal((X.shape[0], 20)))
t = time.perf_counter(); out = eng.xt_matmul_f64(Wd, mean=mean, std=np.sqrt(np.where(var < 1e-12, 1e-12, var))); t2 = time.perf_counter() - t
print("gene statistics %.3f s, z-scored OLS product (k = 20) %.3f s on the stored entries; resident: %s"
      % (t1, t2, [a for a, b in eng.matrix_images().items() if b]), flush=True)
