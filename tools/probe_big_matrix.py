"""Maximum-size probe (round 6): a matrix whose padded element count exceeds 2^31 (default 1 100 000 cells x 2 000 genes:
2.25e9 padded elements, count planes of 4.5 GB each) through the default f16 count path at 1024 packed columns, against the SAME
restarts on the exact-f32 matrix pipe at 32 packed columns (other kernels, other index arithmetic): a 32-bit overflow in either
would show as a mismatch or a fault.  Needs ~25 GB of host memory (checked first) and ~40 GB on the device.
    python tools/probe_big_matrix.py [n_cells]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

n_cells = int(sys.argv[1]) if len(sys.argv) > 1 else 1_100_000
G, K = 2000, 9
avail = [int(l.split()[1]) for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0] / 1e6
need = 3.0 * n_cells * G * 4 / 1e9
print("host memory available %.0f GB, needed ~%.0f GB" % (avail, need), flush=True)
if avail < need + 8:
    print("SKIPPED: not enough host memory for this probe")
    sys.exit(0)
from cnmf_amd.engine import Engine
from oracle import nmf_cd

rs = np.random.RandomState(5)
Hgt = rs.gamma(0.3, 1.0, size=(K, G)); Hgt /= Hgt.sum(axis=1, keepdims=True)
X = np.empty((n_cells, G), dtype=np.float32)
t0 = time.time()
for s in range(0, n_cells, 50_000):
    e = min(s + 50_000, n_cells)
    U = rs.dirichlet(0.3 * np.ones(K), size=e - s) * rs.lognormal(7.0, 0.3, size=(e - s, 1))
    X[s:e] = rs.poisson(U @ Hgt)
sd = X.std(axis=0, ddof=1, dtype=np.float64); sd[sd == 0] = 1.0
X /= sd.astype(np.float32)
print("matrix %d x %d built in %.0f s" % (n_cells, G, time.time() - t0), flush=True)
eng = Engine(0)
eng.set_matrix(X)
ks = [9] * 100 + [13] * 10          # 1 030 columns: a 1024-wide batch with a queue
seeds = list(range(7, 7 + len(ks)))
t0 = time.time()
H, W, n, v = eng.nmf_batch(ks, seeds=seeds, max_iter=6, tol=0.0, warn=False, return_W=False)
st = dict(eng.last_stats)
print("default path: kc %d gemm_mode %d, 6 iterations of %d restarts in %.1f s" % (st["kc"], st["gemm_mode"], len(ks), time.time() - t0), flush=True)
assert st["kc"] == 1024 and st["gemm_mode"] == 4, st
sel = [0, 57, 99, 105]
H32, _, n32, _ = eng.nmf_batch([ks[i] for i in sel], seeds=[seeds[i] for i in sel], max_iter=6, tol=0.0, warn=False, kc_max=32)
st32 = dict(eng.last_stats)
print("reference path: kc %d gemm_mode %d" % (st32["kc"], st32["gemm_mode"]), flush=True)
worst = 0.0
for j, i in enumerate(sel):
    maxabs, relfro = nmf_cd.spectra_error(H32[j].astype(np.float64), H[i])
    worst = max(worst, relfro)
    assert maxabs <= 1e-4 and relfro <= 1e-3, (i, maxabs, relfro)
# the LAST cells of the matrix take part: a usage refit of the first restart's spectra, checked on the tail rows in float64
Hn = H[0] / H[0].sum(axis=1, keepdims=True)
Wd, _ = eng.nnls(Hn, max_iter=30, warn=False)
tail = slice(n_cells - 3000, n_cells)
W_ref, _ = nmf_cd.nnls(X[tail].astype(np.float64), Hn.astype(np.float64), max_iter=30)
err = np.abs(Wd[tail] - W_ref).max() / np.abs(W_ref).max()
assert err <= 2e-3, err
print("OK: f16 count path at 1024 columns == exact-f32 pipe at 32 columns on %d x %d (padded elements %.3g > 2^31 = %s): worst "
      "relative Frobenius %.2e; usage refit of the last 3000 cells vs float64: %.2e"
      % (n_cells, G, float((-(-n_cells // 256) * 256)) * 2048, (-(-n_cells // 256) * 256) * 2048 > 2**31, worst, err))
