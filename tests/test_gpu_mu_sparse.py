"""Kullback-Leibler multiplicative updates on the NON-ZEROS of X (kernels_mu_sparse.hip.h) through the C-ABI: what
scikit-learn computes for scipy.sparse input (_nmf.py:84-194, 526-728) and what cNMF hands it whenever the normalised
counts are stored sparse (cnmf.py:618-631).  Against the float64 oracle (same mathematics: the quotient vanishes where X
does), against the dense matrix-pipe path of the same library, batch independence and run-to-run identity bit for bit,
the refit, the path selection by density, and shapes that need one / several blocks of the other side on each half-step."""
import numpy as np
import pytest

from cnmf_amd import synth
from oracle import nmf_cd, nmf_mu

pytestmark = pytest.mark.gpu


def _sparse_counts(n, g, mu_lib, seed):
    C, _ = synth.topic_counts(n, g, 6, mu_lib, 0.4, seed)
    return synth.normalise_like_prepare(C, dtype=np.float32)


@pytest.mark.parametrize("n,g,k", [(3000, 700, 7),       # padded rank 16: genes one block, cells two
                                   (1500, 2300, 12),      # genes two blocks, cells one
                                   (1200, 900, 16),       # all four quads of a 16-float row live (four gathers in flight)
                                   (2500, 1300, 24),      # padded rank 32 (blocks of 1024): three and two blocks
                                   (130, 70, 3)])         # below one slice group
def test_kl_on_the_non_zeros_vs_oracle_and_dense_path(engine, monkeypatch, n, g, k):
    X = _sparse_counts(n, g, 3.0 if g < 100 else (4.6 if g < 1000 else 5.2), seed=n)
    n, g = X.shape
    auto = (X != 0).mean() < 0.20                                 # (below a quarter the library takes the path by itself)
    engine.set_matrix(X)
    ks, seeds = [k, max(2, k - 2), k], [11, 12, 13]
    if not auto:
        monkeypatch.setenv("CNMF_MU_SPARSE", "1")
    H, W, n_iter, err = engine.nmf_mu_batch(ks, seeds=seeds, max_iter=150, return_W=True, warn=False)
    monkeypatch.setenv("CNMF_MU_SPARSE", "1")
    H1, W1, n1, err1 = engine.nmf_mu_batch(ks, seeds=seeds, max_iter=150, return_W=True, warn=False)
    monkeypatch.setenv("CNMF_MU_SPARSE", "0")
    Hd, Wd, nd, errd = engine.nmf_mu_batch(ks, seeds=seeds, max_iter=150, return_W=True, warn=False)
    monkeypatch.delenv("CNMF_MU_SPARSE")
    for i in range(3):                                            # chosen by density == forced; run to run identical
        np.testing.assert_array_equal(H[i], H1[i]); np.testing.assert_array_equal(W[i], W1[i])
    assert list(n_iter) == list(n1) and list(err) == list(err1)
    diff_dense = any(not np.array_equal(H[i], Hd[i]) for i in range(3))
    assert diff_dense                                             # (another summation order: the dense path really is another path)
    X64 = X.astype(np.float64)
    for i in range(3):
        W_ref, H_ref, n_ref = nmf_mu.nmf_mu(X64, ks[i], seed=seeds[i], max_iter=150)
        assert abs(int(n_iter[i]) - n_ref) <= 10, (i, n_iter[i], n_ref)
        if int(n_iter[i]) != n_ref:
            W_ref, H_ref, _ = nmf_mu.nmf_mu(X64, ks[i], seed=seeds[i], max_iter=int(n_iter[i]), tol=0.0)
        maxabs, relfro = nmf_cd.spectra_error(H_ref, H[i])
        assert maxabs <= 1e-4 and relfro <= 1e-3, (i, maxabs, relfro)
        assert np.abs(W[i] - W_ref).max() <= 2e-3 * np.abs(W_ref).max()
        ref_err = nmf_mu.beta_divergence(X64, W_ref, H_ref, 1, square_root=True)
        assert abs(err[i] - ref_err) <= 1e-3 * ref_err
        if int(nd[i]) == int(n_iter[i]):                          # (pytest -s: this path and the dense matrix-pipe path vs float64)
            print("k=%d: relative Frobenius error vs float64: non-zero path %.2e, dense path %.2e"
                  % (ks[i], relfro, nmf_cd.spectra_error(H_ref, Hd[i])[1]))
    # a restart's result does not depend on what else is in the batch
    Hs, Ws, ns, _ = engine.nmf_mu_batch([ks[1]], seeds=[seeds[1]], max_iter=150, return_W=True, warn=False)
    assert int(ns[0]) == int(n_iter[1])
    np.testing.assert_array_equal(Hs[0], H[1]); np.testing.assert_array_equal(Ws[0], W[1])


def test_kl_non_zero_path_refit_regularisation_and_custom_init(engine, monkeypatch):
    X = _sparse_counts(2600, 900, 4.8, seed=5)
    X64 = X.astype(np.float64)
    engine.set_matrix(X)
    monkeypatch.setenv("CNMF_MU_SPARSE", "1")
    _, Hr, _ = nmf_mu.nmf_mu(X64, 5, seed=1, max_iter=60)
    Hn = Hr / Hr.sum(axis=1, keepdims=True)
    W_ref, n_ref = nmf_mu.nnls_mu(X64, Hn, max_iter=300)
    W, n = engine.nnls_mu(Hn, max_iter=300, warn=False)
    assert abs(n - n_ref) <= 10
    if n != n_ref:
        W_ref, _ = nmf_mu.nnls_mu(X64, Hn, max_iter=int(n), tol=0.0)
    assert np.abs(W - W_ref).max() <= 1e-3 * np.abs(W_ref).max()
    W0, H0 = nmf_cd.random_init(X64, 6, 11)
    W_ref, H_ref, n_ref = nmf_mu.nmf_mu(X64, 6, W0=W0, H0=H0, max_iter=200, alpha_W=0.001, alpha_H=0.002, l1_ratio=0.5)
    H, _, n_iter, _ = engine.nmf_mu_batch([6], W0=[W0], H0=[H0], max_iter=200, alpha_W=0.001, alpha_H=0.002,
                                          l1_ratio=0.5, warn=False)
    assert abs(int(n_iter[0]) - n_ref) <= 10
    if int(n_iter[0]) != n_ref:
        W_ref, H_ref, _ = nmf_mu.nmf_mu(X64, 6, W0=W0, H0=H0, max_iter=int(n_iter[0]), tol=0.0, alpha_W=0.001,
                                        alpha_H=0.002, l1_ratio=0.5)
    maxabs, relfro = nmf_cd.spectra_error(H_ref, H[0])
    assert maxabs <= 1e-4 and relfro <= 1e-3, (maxabs, relfro)


def test_kl_path_selection_and_matrix_change(engine, monkeypatch):
    """A mostly non-zero matrix keeps the dense kernels (same bits as CNMF_MU_SPARSE=0); a new matrix drops the images of
    the old one; ranks above 32 and Itakura-Saito never take the non-zero path."""
    Xd = synth.make_config("C1", dtype=np.float32, n_cells=600)               # ~60 % non-zero
    assert (Xd != 0).mean() > 0.25
    engine.set_matrix(Xd)
    H, _, n, _ = engine.nmf_mu_batch([5, 9], seeds=[3, 4], max_iter=40, warn=False)
    monkeypatch.setenv("CNMF_MU_SPARSE", "0")
    H0, _, n0, _ = engine.nmf_mu_batch([5, 9], seeds=[3, 4], max_iter=40, warn=False)
    monkeypatch.setenv("CNMF_MU_SPARSE", "1")                                 # forced: allowed on any matrix
    H1, _, n1, _ = engine.nmf_mu_batch([5, 9], seeds=[3, 4], max_iter=40, warn=False)
    Hb, _, nb, _ = engine.nmf_mu_batch([40], seeds=[3], max_iter=20, warn=False)                 # rank 40: dense kernels
    monkeypatch.setenv("CNMF_MU_SPARSE", "0")
    Hb0, _, _, _ = engine.nmf_mu_batch([40], seeds=[3], max_iter=20, warn=False)
    # Itakura-Saito needs a strictly positive matrix (scikit-learn's rule, mirrored by the engine)
    engine.set_matrix(Xd + np.float32(1e-3))
    Hi0, _, _, _ = engine.nmf_mu_batch([5], seeds=[3], beta_loss="itakura-saito", max_iter=20, warn=False)
    monkeypatch.setenv("CNMF_MU_SPARSE", "1")
    Hi, _, ni, _ = engine.nmf_mu_batch([5], seeds=[3], beta_loss="itakura-saito", max_iter=20, warn=False)
    monkeypatch.delenv("CNMF_MU_SPARSE")
    for a, b in zip(H, H0):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(Hb[0], Hb0[0]); np.testing.assert_array_equal(Hi[0], Hi0[0])
    for a, b in zip(H, H1):
        assert not np.array_equal(a, b)
        _, rel = nmf_cd.spectra_error(a.astype(np.float64), b)
        assert rel <= 1e-3
    # another matrix: the images are rebuilt for it
    Xs = _sparse_counts(1000, 400, 4.6, seed=9)
    engine.set_matrix(Xs)
    Hs, _, ns, _ = engine.nmf_mu_batch([4], seeds=[8], max_iter=80, warn=False)
    _, H_ref, _ = nmf_mu.nmf_mu(Xs.astype(np.float64), 4, seed=8, max_iter=int(ns[0]), tol=0.0)
    maxabs, relfro = nmf_cd.spectra_error(H_ref, Hs[0])
    assert maxabs <= 1e-4 and relfro <= 1e-3


def test_kl_non_zero_path_through_the_cnmf_callsite(engine, tmp_path):
    """`cNMF.factorize` with beta_loss='kullback-leibler' on a count matrix at a real matrix's density: the engine counts the
    non-zeros of what it was handed and takes the non-zero path by itself; the merged spectra equal the oracle's restarts."""
    from cnmf_amd.cnmf import cNMF, ledger_seeds
    X = _sparse_counts(1800, 600, 4.6, seed=21)
    assert (X != 0).mean() < 0.25
    obj = cNMF(output_dir=str(tmp_path), name="mu_sparse", engine=engine)
    obj.prepare_from_matrix(X, components=[4, 6], n_iter=2, seed=14, beta_loss="kullback-leibler", max_NMF_iter=120)
    obj.factorize()
    led = ledger_seeds([4, 6], 2, 14)
    X64 = X.astype(np.float64)
    for k in (4, 6):
        merged = obj.combine_nmf(k)
        assert merged.shape == (2 * k, X.shape[1]) and np.isfinite(merged.values).all() and (merged.values >= 0).all()
        seed = [row[2] for row in led if row[0] == k][0]
        _, H_ref, _ = nmf_mu.nmf_mu(X64, k, seed=seed, max_iter=120)
        maxabs, relfro = nmf_cd.spectra_error(H_ref, merged.values[:k])
        assert maxabs <= 1e-4 and relfro <= 1e-3, (k, maxabs, relfro)
