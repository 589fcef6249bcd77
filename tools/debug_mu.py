import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cnmf_amd import synth
from cnmf_amd.engine import Engine
from oracle import nmf_cd, nmf_mu
X = synth.make_config("C1", dtype=np.float64, n_cells=700)
eng = Engine(0); eng.set_matrix(X)
for k, seed, mi in ((7, 59886188, 1), (7, 59886188, 2), (7, 59886188, 10), (7, 59886188, 400), (20, 9, 3)):
    W_ref, H_ref, n_ref = nmf_mu.nmf_mu(X, k, seed=seed, max_iter=mi)
    H, W, n_iter, err = eng.nmf_mu_batch([k], seeds=[seed], max_iter=mi, return_W=True, warn=False)
    print(k, mi, "n", n_iter, n_ref, "H rel", np.abs(H[0] - H_ref).max() / np.abs(H_ref).max(), "W rel", np.abs(W[0] - W_ref).max() / np.abs(W_ref).max(),
          "err", err[0], nmf_mu.beta_divergence(X, W_ref, H_ref, 1, True), flush=True)
    if mi <= 2:
        d = np.abs(W[0] - W_ref) / np.abs(W_ref).max()
        i = np.unravel_index(np.argmax(d), d.shape); print("  worst W at", i, W[0][i], W_ref[i], "rows bad:", np.where(d.max(axis=1) > 1e-3)[0][:20])
        d = np.abs(H[0] - H_ref) / np.abs(H_ref).max()
        i = np.unravel_index(np.argmax(d), d.shape); print("  worst H at", i, H[0][i], H_ref[i], "genes bad:", np.where(d.max(axis=0) > 1e-3)[0][:20])
