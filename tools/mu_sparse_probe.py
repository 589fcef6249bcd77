"""Kullback-Leibler restarts on the non-zeros (kernels_mu_sparse.hip.h) against the dense matrix-pipe kernels: time per
restart-iteration at 200 000 x 2 000 cells x genes (BASELINE config 4's shape) with the library size of a real 10x matrix
(~9 % non-zero) and with the bench's own synthetic C4 (34 %); CNMF_DEBUG=1 prints the padding of the images."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cnmf_amd import synth
from cnmf_amd.engine import Engine

n_cells = int(os.environ.get("SP_CELLS", "200000"))
its = int(os.environ.get("MU_ITERS", "30"))
eng = Engine(0)
cases = (("library size e^5.2", 5.2), ("library size e^6.8 (bench C4)", 6.8))
modes = tuple(os.environ.get("SP_MODES", "1,0").split(","))
batches = ([9], [9] * 8, [9] * 32, [5, 6, 7, 8, 9, 10, 11, 12, 13] * 4, [20] * 32)
if os.environ.get("SP_ONLY"):                     # (kernel traces / counters: the real-density matrix only)
    cases = cases[:1]
    if "SP_MODES" not in os.environ:
        modes = ("1",)
if os.environ.get("SP_KS"):                       # e.g. SP_KS=13x32,16x32
    batches = tuple([int(a)] * int(b) for a, b in (t.split("x") for t in os.environ["SP_KS"].split(",")))
elif os.environ.get("SP_LONG"):                     # (steady state: the per-call set-up amortised like in a real run)
    batches = ([9] * 32, [5, 6, 7, 8, 9, 10, 11, 12, 13] * 4, [20] * 32)
for label, mu_lib in cases:
    C, _ = synth.topic_counts(n_cells, 2000, 20, mu_lib, 0.4, 3)
    X = synth.normalise_like_prepare(C, dtype=np.float32)
    del C
    print("%s: %d x %d, %.1f %% non-zero" % (label, X.shape[0], X.shape[1], 100.0 * (X != 0).mean()), flush=True)
    for mode in modes:
        eng.set_matrix(X)                        # (drops the images: CNMF_SP_ORDER / CNMF_MU_SPARSE are read when they are built)
        os.environ["CNMF_MU_SPARSE"] = mode
        t = time.perf_counter()
        eng.nmf_mu_batch([5, 20], seeds=[1, 2], max_iter=3, tol=0, warn=False)          # warm up: images / X^T, code objects
        print("  CNMF_MU_SPARSE=%s: first call (images) %.2f s" % (mode, time.perf_counter() - t), flush=True)
        for ks in batches:
            t = time.perf_counter()
            H, _, n, err = eng.nmf_mu_batch(ks, seeds=list(range(7, 7 + len(ks))), max_iter=its, tol=0, warn=False)
            dt = time.perf_counter() - t
            print("  CNMF_MU_SPARSE=%s KL k=%s x%d: %d iterations each in %.3f s -> %.1f us per restart-iteration, err %.6g"
                  % (mode, sorted(set(ks)), len(ks), n[0], dt, dt / n.sum() * 1e6, err[0]), flush=True)
    del X
