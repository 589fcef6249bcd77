"""GPU parity of the multiplicative-update solver (beta_loss = kullback-leibler / itakura-saito),
the reference's path for beta_loss != 'frobenius' (cnmf.py:618-631), against the numpy
restatement of sklearn's _fit_multiplicative_update (oracle/nmf_mu.py, pinned to sklearn).

Tolerance: multiplicative updates are smooth, so the fp32 device trajectory tracks the float64
oracle closely: normalised spectra max-abs <= 1e-4, rel-Frobenius <= 1e-3; n_iter (a multiple of
10, the divergence is evaluated every 10 iterations) within one evaluation period."""
import numpy as np
import pytest

from cnmf_amd import synth
from oracle import nmf_cd, nmf_mu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def X():
    return synth.make_config("C1", dtype=np.float64, n_cells=700)


@pytest.mark.parametrize("beta_loss,k,seed", [("kullback-leibler", 7, 59886188), ("kullback-leibler", 12, 5),
                                              ("itakura-saito", 5, 3), ("kullback-leibler", 20, 9)])
def test_mu_restart_vs_oracle(engine, X, beta_loss, k, seed):
    Xp = X + (1e-3 if beta_loss == "itakura-saito" else 0.0)
    engine.set_matrix(Xp)
    W_ref, H_ref, n_ref = nmf_mu.nmf_mu(Xp, k, seed=seed, beta_loss=beta_loss, max_iter=400)
    H, W, n_iter, err = engine.nmf_mu_batch([k], seeds=[seed], beta_loss=beta_loss, max_iter=400,
                                            return_W=True, warn=False)
    assert abs(int(n_iter[0]) - n_ref) <= 10, (n_iter, n_ref)
    if int(n_iter[0]) != n_ref:       # one evaluation period apart: compare at the device's own truncation
        W_ref, H_ref, _ = nmf_mu.nmf_mu(Xp, k, seed=seed, beta_loss=beta_loss, max_iter=int(n_iter[0]), tol=0.0)
    maxabs, relfro = nmf_cd.spectra_error(H_ref, H[0])
    assert maxabs <= 1e-4 and relfro <= 1e-3, (maxabs, relfro)
    assert np.abs(W[0] - W_ref).max() <= 2e-3 * np.abs(W_ref).max()
    ref_err = nmf_mu.beta_divergence(Xp, W_ref, H_ref, nmf_mu.BETA[beta_loss], square_root=True)
    assert abs(err[0] - ref_err) <= 2e-3 * ref_err


def test_mu_custom_init_and_regularisation(engine, X):
    engine.set_matrix(X)
    W0, H0 = nmf_cd.random_init(X, 6, 11)
    W_ref, H_ref, n_ref = nmf_mu.nmf_mu(X, 6, W0=W0, H0=H0, max_iter=200, alpha_W=0.001, alpha_H=0.002, l1_ratio=0.5)
    H, _, n_iter, _ = engine.nmf_mu_batch([6], W0=[W0], H0=[H0], max_iter=200, alpha_W=0.001, alpha_H=0.002,
                                          l1_ratio=0.5, warn=False)
    assert abs(int(n_iter[0]) - n_ref) <= 10
    if int(n_iter[0]) != n_ref:
        W_ref, H_ref, _ = nmf_mu.nmf_mu(X, 6, W0=W0, H0=H0, max_iter=int(n_iter[0]), tol=0.0, alpha_W=0.001,
                                        alpha_H=0.002, l1_ratio=0.5)
    maxabs, relfro = nmf_cd.spectra_error(H_ref, H[0])
    assert maxabs <= 1e-4 and relfro <= 1e-3, (maxabs, relfro)


def test_mu_refit(engine, X):
    engine.set_matrix(X)
    _, H, _ = nmf_mu.nmf_mu(X, 5, seed=1, max_iter=60)
    Hn = H / H.sum(axis=1, keepdims=True)
    W_ref, n_ref = nmf_mu.nnls_mu(X, Hn, max_iter=300)
    W, n = engine.nnls_mu(Hn, max_iter=300, warn=False)
    assert abs(n - n_ref) <= 10
    if n != n_ref:
        W_ref, _ = nmf_mu.nnls_mu(X, Hn, max_iter=int(n), tol=0.0)
    assert np.abs(W - W_ref).max() <= 2e-3 * np.abs(W_ref).max()


def test_mu_through_cnmf_callsite(engine, X, tmp_path):
    from cnmf_amd.cnmf import cNMF
    obj = cNMF(output_dir=str(tmp_path), name="mu", engine=engine)
    obj.prepare_from_matrix(X, components=[4], n_iter=3, seed=14, beta_loss="kullback-leibler", max_NMF_iter=120)
    obj.factorize()
    merged = obj.combine_nmf(4)
    assert merged.shape == (12, X.shape[1]) and np.isfinite(merged.values).all() and (merged.values >= 0).all()
    import yaml
    kw = yaml.load(open(obj.paths["nmf_run_parameters"]), Loader=yaml.FullLoader)
    assert kw["solver"] == "mu" and kw["beta_loss"] == "kullback-leibler"       # as the reference writes it
    from cnmf_amd.cnmf import ledger_seeds
    led = ledger_seeds([4], 3, 14)
    _, H_ref, _ = nmf_mu.nmf_mu(X, 4, seed=led[0][2], max_iter=120)
    maxabs, relfro = nmf_cd.spectra_error(H_ref, merged.values[:4])
    assert maxabs <= 1e-4 and relfro <= 1e-3


def test_mu_batch_matches_single_and_valu_path(engine, X, monkeypatch):
    """Kullback-Leibler restarts run batched on the matrix pipe (kernels_mu_mfma.hip.h): a restart's result must not
    depend on what else is in the batch (bit for bit), and must agree with the vector-ALU path and the oracle."""
    engine.set_matrix(X)
    ks = [5, 7, 12, 20, 9, 16, 17, 3, 8, 13, 6, 11, 10, 4, 15, 14, 19, 2]          # more than one round of 16 slots
    seeds = [100 + i for i in range(len(ks))]
    H, W, n_iter, err = engine.nmf_mu_batch(ks, seeds=seeds, max_iter=200, return_W=True, warn=False)
    for i in (0, 3, 6, 17):
        H1, W1, n1, e1 = engine.nmf_mu_batch([ks[i]], seeds=[seeds[i]], max_iter=200, return_W=True, warn=False)
        assert int(n1[0]) == int(n_iter[i])
        np.testing.assert_array_equal(H1[0], H[i])
        np.testing.assert_array_equal(W1[0], W[i])
    for i in (1, 2, 16):
        W_ref, H_ref, _ = nmf_mu.nmf_mu(X, ks[i], seed=seeds[i], max_iter=int(n_iter[i]), tol=0.0)
        maxabs, relfro = nmf_cd.spectra_error(H_ref, H[i])
        assert maxabs <= 1e-4 and relfro <= 1e-3, (i, maxabs, relfro)
        ref_err = nmf_mu.beta_divergence(X, W_ref, H_ref, 1, square_root=True)
        assert abs(err[i] - ref_err) <= 2e-3 * ref_err
    # the vector-ALU path (float32 FMAs in another order): two float32 trajectories, held to a looser bound
    monkeypatch.setenv("CNMF_MU_VALU", "1")
    Hv, Wv, nv, ev = engine.nmf_mu_batch(ks[:6], seeds=seeds[:6], max_iter=200, return_W=True, warn=False)
    monkeypatch.delenv("CNMF_MU_VALU")
    for i in range(6):
        assert abs(int(nv[i]) - int(n_iter[i])) <= 10
        if int(nv[i]) == int(n_iter[i]):
            maxabs, relfro = nmf_cd.spectra_error(Hv[i], H[i])
            assert maxabs <= 2e-3 and relfro <= 2e-3, (i, maxabs, relfro)


def test_itakura_saito_batch_vs_oracle(engine, X):
    """Itakura-Saito restarts run on the same batched matrix-pipe kernels (a second ratio and a second set of
    accumulators for the denominator): ranks on both register layouts, more restarts than one round of launches
    needs, each against the float64 oracle; and batch independence bit for bit."""
    Xp = X + 1e-3
    engine.set_matrix(Xp)
    ks = [5, 3, 16, 17, 9, 24, 7]
    seeds = [40 + i for i in range(len(ks))]
    H, W, n_iter, err = engine.nmf_mu_batch(ks, seeds=seeds, beta_loss="itakura-saito", max_iter=120, return_W=True, warn=False)
    for i in (0, 2, 3, 5):
        W_ref, H_ref, n_ref = nmf_mu.nmf_mu(Xp, ks[i], seed=seeds[i], beta_loss="itakura-saito", max_iter=120)
        assert abs(int(n_iter[i]) - n_ref) <= 10, (i, n_iter[i], n_ref)
        if int(n_iter[i]) != n_ref:
            W_ref, H_ref, _ = nmf_mu.nmf_mu(Xp, ks[i], seed=seeds[i], beta_loss="itakura-saito", max_iter=int(n_iter[i]), tol=0.0)
        maxabs, relfro = nmf_cd.spectra_error(H_ref, H[i])
        assert maxabs <= 1e-4 and relfro <= 1e-3, (i, maxabs, relfro)
        assert np.abs(W[i] - W_ref).max() <= 2e-3 * np.abs(W_ref).max()
        ref_err = nmf_mu.beta_divergence(Xp, W_ref, H_ref, 0, square_root=True)
        assert abs(err[i] - ref_err) <= 2e-3 * ref_err
    H1, W1, n1, _ = engine.nmf_mu_batch([ks[3]], seeds=[seeds[3]], beta_loss="itakura-saito", max_iter=120, return_W=True, warn=False)
    assert int(n1[0]) == int(n_iter[3])
    np.testing.assert_array_equal(H1[0], H[3])
    np.testing.assert_array_equal(W1[0], W[3])


def test_mu_ranks_33_to_64_batched_on_the_matrix_pipe(engine, X, monkeypatch):
    """Ranks 33..64 run on the matrix-pipe kernels at padded rank 64 (two restarts per workgroup, two M tiles of the
    second product): an odd number of restarts (the last workgroup holds one), results independent of the batch bit
    for bit, against the oracle, and against the vector-ALU kernels that served these ranks before."""
    engine.set_matrix(X)
    ks = [40, 33, 64, 48, 57]
    seeds = [300 + i for i in range(len(ks))]
    H, W, n_iter, err = engine.nmf_mu_batch(ks, seeds=seeds, max_iter=60, return_W=True, warn=False)
    for i in (0, 2, 4):
        H1, W1, n1, _ = engine.nmf_mu_batch([ks[i]], seeds=[seeds[i]], max_iter=60, return_W=True, warn=False)
        assert int(n1[0]) == int(n_iter[i])
        np.testing.assert_array_equal(H1[0], H[i])
        np.testing.assert_array_equal(W1[0], W[i])
    for i in (1, 2, 3):
        W_ref, H_ref, _ = nmf_mu.nmf_mu(X, ks[i], seed=seeds[i], max_iter=int(n_iter[i]), tol=0.0)
        R, R_ref = W[i].astype(np.float64) @ H[i], W_ref @ H_ref
        assert np.abs(R - R_ref).max() <= 2e-3 * np.abs(R_ref).max(), ks[i]
        ref_err = nmf_mu.beta_divergence(X, W_ref, H_ref, 1, square_root=True)
        assert abs(err[i] - ref_err) <= 2e-3 * ref_err
    monkeypatch.setenv("CNMF_MU_VALU", "1")
    Hv, Wv, nv, ev = engine.nmf_mu_batch(ks[:2], seeds=seeds[:2], max_iter=60, return_W=True, warn=False)
    monkeypatch.delenv("CNMF_MU_VALU")
    for i in range(2):
        assert abs(int(nv[i]) - int(n_iter[i])) <= 10
        if int(nv[i]) == int(n_iter[i]):
            R, Rv = W[i].astype(np.float64) @ H[i], Wv[i].astype(np.float64) @ Hv[i]
            assert np.abs(R - Rv).max() <= 3e-3 * np.abs(Rv).max()
            assert abs(err[i] - ev[i]) <= 2e-3 * ev[i]


def test_mu_at_scale_vs_oracle(engine):
    """12 000 cells x 2 000 genes (375 32-cell strips, 16 gene blocks, 94 cell blocks of 128): Kullback-Leibler at ranks
    9, 20, 40 and 64 -- all three register layouts of the matrix-pipe kernels -- and Itakura-Saito at rank 40, a fixed
    number of iterations against the float64 oracle."""
    X = synth.make_config("C3", dtype=np.float64, n_cells=12000)
    engine.set_matrix(X)
    ks, seeds = [9, 20, 40, 64], [5, 6, 7, 8]
    H, W, n_iter, err = engine.nmf_mu_batch(ks, seeds=seeds, max_iter=12, tol=0.0, return_W=True, warn=False)
    for k, seed, h, w, n, e in zip(ks, seeds, H, W, n_iter, err):
        W_ref, H_ref, _ = nmf_mu.nmf_mu(X, k, seed=seed, max_iter=12, tol=0.0)
        assert int(n) == 12
        maxabs, relfro = nmf_cd.spectra_error(H_ref, h)
        assert maxabs <= 1e-4 and relfro <= 1e-3, (k, maxabs, relfro)
        assert np.abs(w - W_ref).max() <= 2e-3 * np.abs(W_ref).max(), k
        ref_err = nmf_mu.beta_divergence(X, W_ref, H_ref, 1, square_root=True)
        assert abs(e - ref_err) <= 2e-3 * ref_err, (k, e, ref_err)
    Xp = X + 1e-3
    engine.set_matrix(Xp)
    Hi, Wi, ni, ei = engine.nmf_mu_batch([40], seeds=[9], beta_loss="itakura-saito", max_iter=8, tol=0.0, return_W=True, warn=False)
    W_ref, H_ref, _ = nmf_mu.nmf_mu(Xp, 40, seed=9, beta_loss="itakura-saito", max_iter=8, tol=0.0)
    R, R_ref = Wi[0].astype(np.float64) @ Hi[0], W_ref @ H_ref
    assert np.abs(R - R_ref).max() <= 5e-3 * np.abs(R_ref).max()
