"""Kullback-Leibler end to end (round 5): the float64 refit on the stored entries (cnmf_mu_refit_f64, csr_host.hip.h) against
the float64 oracle, and the consensus tail of a Kullback-Leibler run against the artefacts the UNMODIFIED reference wrote
under ``beta_loss='kullback-leibler'`` (tests/golden/ref_small_kl.npz, tools/make_golden.py --beta-loss kullback-leibler) at
the reference's own bar -- sum of squared differences < 1e-4 (/root/reference/tests/test_reproducibility.py:96-115) --
through the C-ABI and through the cNMF mirror class, TPM dense and CSR."""
import os

import numpy as np
import pandas as pd
import pytest
import scipy.sparse as sp

from cnmf_amd import synth
from cnmf_amd.cnmf import cNMF, load_df_from_npz, save_df_to_npz
from oracle import consensus as oc
from oracle import nmf_cd, nmf_mu

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "ref_small_kl.npz")
TOLERANCE = 1e-4


@pytest.fixture(scope="module")
def g():
    return dict(np.load(GOLD, allow_pickle=False))


def _counts(n, g_, mu_lib, seed):
    C, _ = synth.topic_counts(n, g_, 6, mu_lib, 0.4, seed)
    return synth.normalise_like_prepare(C, dtype=np.float32)


def _oracle_refit(X64, H, max_iter, **kw):
    return nmf_mu.nnls_mu(X64, H, max_iter=max_iter, **kw)


@pytest.mark.parametrize("n,g_,k", [(2600, 900, 5),      # padded rank 8
                                    (1500, 2300, 12),    # 16
                                    (1200, 700, 20),     # 32
                                    (900, 500, 40)])     # 64: the factor row read twice
def test_mu_refit_f64_vs_oracle(engine, n, g_, k):
    """refit_usage and refit_spectra with solver='mu' / Kullback-Leibler: the same iteration count as the float64 oracle
    (pinned to scikit-learn, dense and scipy.sparse input: tests/test_oracle_mu.py) and agreement to round-off -- on a
    dense upload (compressed rows built on the device), on a CSR upload (arrays kept), and on a CSR upload that is not in
    canonical form (falls back to the rebuilt rows): bit-identical between the three."""
    X = _counts(n, g_, 4.8, seed=n)
    n, g_ = X.shape
    X64 = X.astype(np.float64)
    rs = np.random.RandomState(k)
    H = np.abs(rs.standard_normal((k, g_))) * (rs.rand(k, g_) < 0.5)
    H /= H.sum(axis=1, keepdims=True)
    U = np.abs(rs.standard_normal((n, k)))
    U /= U.sum(axis=1, keepdims=True)
    W_ref, n_ref = _oracle_refit(X64, H, 300)
    Wt_ref, nt_ref = _oracle_refit(np.ascontiguousarray(X64.T), np.ascontiguousarray(U.T), 300)
    res = []
    Xs = sp.csr_matrix(X)
    for label, M in (("dense", X), ("csr", Xs)):
        engine.set_matrix(M)
        W, it, err = engine.mu_refit_f64(H, max_iter=300, warn=False)
        Wt, itt, errt = engine.mu_refit_f64(U.T, transposed=True, max_iter=300, warn=False)
        assert W.dtype == np.float64 and it == n_ref and itt == nt_ref, (label, it, n_ref, itt, nt_ref)
        assert np.abs(W - W_ref).max() <= 1e-9 * np.abs(W_ref).max(), label
        assert np.abs(Wt - Wt_ref).max() <= 1e-9 * np.abs(Wt_ref).max(), label
        ref_err = nmf_mu.beta_divergence(X64, W_ref, H, 1, square_root=True)
        assert abs(err - ref_err) <= 1e-10 * ref_err
        res.append((W, Wt))
    np.testing.assert_array_equal(res[0][0], res[1][0])          # the device-built rows ARE the uploaded ones
    np.testing.assert_array_equal(res[0][1], res[1][1])
    # the transpose of the compressed rows is built on the device by a counting sort whose order is fixed by construction:
    # rebuilt from scratch (a fresh upload each time) it must give the same bits, run after run
    for _ in range(3):
        engine.set_matrix(Xs)
        Wt_again, _, _ = engine.mu_refit_f64(U.T, transposed=True, max_iter=300, warn=False)
        np.testing.assert_array_equal(Wt_again, res[1][1])
    # a CSR upload with unsorted rows and a duplicated entry: the arrays are not kept, same numbers
    w0 = engine.init_scale(k)                                     # (sqrt(X.mean() / k) in X's dtype, as the two runs above)
    coo = Xs.tocoo()
    rows = np.concatenate([coo.row[::-1], coo.row[:1]])
    cols = np.concatenate([coo.col[::-1], coo.col[:1]])
    vals = np.concatenate([coo.data[::-1], coo.data[:1]]).astype(np.float32)
    vals[-1] *= 0.5; vals[len(coo.data) - 1] *= 0.5             # the first entry, split in two halves (exact in float32)
    order = np.argsort(rows, kind="stable")
    indptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=n))]).astype(np.int32)
    import ctypes as C
    ip = C.POINTER(C.c_int32)
    idx32 = np.ascontiguousarray(cols[order], dtype=np.int32)
    v32 = np.ascontiguousarray(vals[order], dtype=np.float32)
    engine._check(engine._lib.cnmf_set_matrix_csr(engine._ctx, indptr.ctypes.data_as(ip), idx32.ctypes.data_as(ip),
                                                  v32.ctypes.data_as(C.POINTER(C.c_float)), n, g_))
    engine.shape = (n, g_)
    np.testing.assert_array_equal(engine.get_matrix(), X)
    W2, it2, _ = engine.mu_refit_f64(H, max_iter=300, warn=False, w_init=w0)
    np.testing.assert_array_equal(W2, res[0][0])


def test_mu_refit_f64_penalties_column_subset_and_stopping_rule(engine):
    X = _counts(1800, 800, 5.0, seed=4)
    X64 = X.astype(np.float64)
    n, g_ = X.shape
    rs = np.random.RandomState(1)
    k = 7
    H = np.abs(rs.standard_normal((k, g_)))
    engine.set_matrix(sp.csr_matrix(X))
    # l1 / l2 penalties on W (scaled by the feature count like scikit-learn, _nmf.py:1254-1265); max_iter not a multiple of 10
    for kw in (dict(alpha_W=0.003, l1_ratio=0.0), dict(alpha_W=0.002, l1_ratio=0.7), dict()):
        for max_iter in (37, 200):
            W_ref, n_ref = _oracle_refit(X64, H, max_iter, **kw)
            W, it, _ = engine.mu_refit_f64(H, max_iter=max_iter, warn=False, **kw)
            assert it == n_ref, (kw, max_iter, it, n_ref)
            assert np.abs(W - W_ref).max() <= 1e-9 * np.abs(W_ref).max(), (kw, max_iter)
    # tol = 0: no stopping rule, every iteration runs
    W_ref, n_ref = _oracle_refit(X64, H, 25, tol=0.0)
    W, it, _ = engine.mu_refit_f64(H, max_iter=25, tol=0.0, warn=False)
    assert it == n_ref == 25 and np.abs(W - W_ref).max() <= 1e-9 * np.abs(W_ref).max()
    # the column subset divided by a per-column constant: tpm[:, hvgs] / std of the final usage refit (cnmf.py:963-972)
    sel = np.sort(rs.choice(g_, 300, replace=False))
    std1 = X64[:, sel].std(axis=0, ddof=1)
    Xsub = X64[:, sel] / std1
    Hs = np.abs(rs.standard_normal((k, 300)))
    W_ref, n_ref = _oracle_refit(Xsub, Hs, 200, alpha_W=0.001)
    div = np.zeros(g_); div[sel] = std1
    H_full = np.zeros((k, g_)); H_full[:, sel] = Hs
    W, it, err = engine.mu_refit_f64(H_full, col_divisor=div, w_init=float(np.sqrt(Xsub.mean() / k)), n_features=300,
                                     max_iter=200, alpha_W=0.001, warn=False)
    assert it == n_ref and np.abs(W - W_ref).max() <= 1e-9 * np.abs(W_ref).max()
    assert abs(err - nmf_mu.beta_divergence(Xsub, W_ref, Hs, 1, square_root=True)) <= 1e-9 * err
    # the refit of the engine's `nnls_mu` call site is this path (float32 result of the float64 solve)
    W32, it32 = engine.nnls_mu(H, max_iter=200, warn=False)
    W64, it64, _ = engine.mu_refit_f64(H, max_iter=200, warn=False)
    assert it32 == it64 and W32.dtype == np.float32 and np.array_equal(W32, W64.astype(np.float32))
    # input validation like scikit-learn's (_nmf.py:1217-1226)
    with pytest.raises(ValueError):
        engine.mu_refit_f64(-H)
    with pytest.raises(ValueError):
        engine.mu_refit_f64(np.zeros_like(H))
    with pytest.raises(ValueError):
        engine.mu_refit_f64(H[:, :-1])
    # a scaling of the resident matrix drops the compressed rows with the other derived images
    engine.set_matrix(X * rs.uniform(0.5, 4.0, size=g_).astype(np.float32))
    Wa, _, _ = engine.mu_refit_f64(H, max_iter=20, warn=False)
    std, _ = engine.scale_genes_unit_variance()
    Wb, _, _ = engine.mu_refit_f64(H, max_iter=20, warn=False)
    Xn = engine.get_matrix().astype(np.float64)
    W_ref, _ = _oracle_refit(Xn, H, 20)
    assert np.abs(Wb - W_ref).max() <= 1e-9 * np.abs(W_ref).max() and np.abs(Wa - Wb).max() > 1e-3 * np.abs(Wb).max()


def _kl_tail_on_device(engine, g, k, thr, sparse):
    X = g["norm_counts"]
    out = engine.consensus(g["merged_k%d" % k], k, density_threshold=thr)
    engine.set_matrix(sp.csr_matrix(X) if sparse else X)
    rf, _, _ = engine.mu_refit_f64(out["median_spectra"], max_iter=1000)                  # cnmf.py:920
    norm = rf / rf.sum(axis=1, keepdims=True)
    order = np.argsort(-norm.sum(axis=0), kind="stable")                                # cnmf.py:939-946
    rf, norm, med = rf[:, order], norm[:, order], out["median_spectra"][order]
    tpm = g["tpm"]
    engine.set_matrix(sp.csr_matrix(tpm) if sparse else tpm)                            # ONE upload for the three steps below
    Wt, _, _ = engine.mu_refit_f64(norm.T, transposed=True, max_iter=1000)              # cnmf.py:952 refit_spectra
    spectra_tpm = Wt.T
    mean, pvar = engine.col_mean_var()
    var = np.where(pvar < 1e-12, 1e-12, pvar)
    XtY = engine.xt_matmul_f64(rf, mean=mean, std=np.sqrt(var))
    coef, *_ = np.linalg.lstsq(rf.T @ rf, XtY, rcond=None)
    hidx = np.array([list(g["tpm_genes"]).index(x) for x in list(g["genes"])])
    n = tpm.shape[0]
    std1 = np.sqrt(pvar[hidx] * n / (n - 1.0))
    srf = spectra_tpm[:, hidx] / g["tpm_stats"][hidx, 1]
    div = np.zeros(tpm.shape[1]); div[hidx] = std1
    H_full = np.zeros((k, tpm.shape[1])); H_full[:, hidx] = srf
    usages, _, _ = engine.mu_refit_f64(H_full, col_divisor=div, w_init=float(np.sqrt((mean[hidx] / std1).mean() / k)),
                                       n_features=len(hidx), max_iter=1000)             # cnmf.py:972
    return med, usages, spectra_tpm, coef


@pytest.mark.parametrize("sparse", [False, True])
@pytest.mark.parametrize("k,thr", [(5, 0.5), (4, 2.0)])
def test_kl_consensus_tail_golden_reference(engine, g, k, thr, sparse):
    med, usages, spectra_tpm, coef = _kl_tail_on_device(engine, g, k, thr, sparse)
    err = {"consensus_spectra": ((med - g["consensus_spectra_k%d" % k]) ** 2).sum(),
           "consensus_usages": ((usages - g["consensus_usages_k%d" % k]) ** 2).sum(),
           "gene_spectra_tpm": ((spectra_tpm - g["gene_spectra_tpm_k%d" % k]) ** 2).sum(),
           "gene_spectra_score": ((coef - g["gene_spectra_score_k%d" % k]) ** 2).sum()}
    print("Kullback-Leibler consensus tail k=%d thr=%s sparse=%s: sum of squared differences vs the reference's files: %s"
          % (k, thr, sparse, {a: float("%.3g" % b) for a, b in err.items()}))
    for name, e in err.items():
        assert e < TOLERANCE, (name, e)


@pytest.mark.parametrize("sparse", [False, True])
def test_mirror_class_kl_consensus_from_reference_merged_spectra(engine, g, tmp_path, sparse):
    """cNMF.consensus / k_selection_stats of the mirror class under beta_loss='kullback-leibler' on the REFERENCE's merged
    spectra, TPM dense or sparse: every artefact the reference pins, at its own tolerance; no host todense() of the TPM
    matrix, no transposed upload (the engine holds ONE matrix at a time: asserted on its shape)."""
    obj = cNMF(output_dir=str(tmp_path), name="goldkl", engine=engine)
    nc = pd.DataFrame(g["norm_counts"], index=["c%d" % i for i in range(g["norm_counts"].shape[0])], columns=list(g["genes"]))
    if sparse:
        tpm = (sp.csr_matrix(g["tpm"]), list(g["tpm_genes"]))
    else:
        tpm = pd.DataFrame(g["tpm"], index=nc.index, columns=list(g["tpm_genes"]))
    obj.prepare_from_matrix(nc, components=[4, 5, 6], n_iter=12, seed=14, beta_loss="kullback-leibler", tpm=tpm)
    import yaml
    kw = yaml.load(open(obj.paths["nmf_run_parameters"]), Loader=yaml.FullLoader)
    assert kw["solver"] == "mu" and kw["beta_loss"] == "kullback-leibler"
    for k in (4, 5, 6):
        idx = ["iter%d_topic%d" % (it, t + 1) for it in range(12) for t in range(k)]
        save_df_to_npz(pd.DataFrame(g["merged_k%d" % k], index=idx, columns=list(g["genes"])), obj.paths["merged_spectra"] % k)
    stats = obj.k_selection_stats()
    for row, k in zip(stats.itertuples(), (4, 5, 6)):
        _, _, sil, err = g["stats_k%d" % k]
        assert row.k == k and abs(row.silhouette - sil) < 1e-8 and abs(row.prediction_error - err) <= 2e-5 * err
    for k, thr in ((5, 0.5), (4, 2.0)):
        med, usages = obj.consensus(k, density_threshold=thr)
        assert engine.shape == g["tpm"].shape                   # the TPM matrix as stored, not a transposed / dense re-upload
        if sparse:
            # the whole tail -- refits, gene statistics, the OLS product with z-scoring -- walked the stored entries: the
            # dense image of the TPM matrix was never formed
            im = engine.matrix_images()
            assert im["csr"] and im["csr_of_transpose"] and not im["dense"] and not im["dense_transpose"], im
        rep = str(thr).replace(".", "_")
        errs = {"consensus_spectra": ((med.values - g["consensus_spectra_k%d" % k]) ** 2).sum(),
                "consensus_usages": ((usages.values - g["consensus_usages_k%d" % k]) ** 2).sum(),
                "gene_spectra_tpm": ((load_df_from_npz(obj.paths["gene_spectra_tpm"] % (k, rep)).values
                                      - g["gene_spectra_tpm_k%d" % k]) ** 2).sum(),
                "gene_spectra_score": ((load_df_from_npz(obj.paths["gene_spectra_score"] % (k, rep)).values
                                        - g["gene_spectra_score_k%d" % k]) ** 2).sum()}
        print("mirror class (Kullback-Leibler) k=%d sparse=%s: %s" % (k, sparse, {a: float("%.3g" % b) for a, b in errs.items()}))
        for name, e in errs.items():
            assert e < TOLERANCE, (name, e)


def test_kl_pipeline_factorize_to_consensus_vs_reference(engine, g, tmp_path):
    """prepare_from_matrix -> factorize -> combine -> k_selection_stats -> consensus under beta_loss='kullback-leibler', the
    restarts on the device's float32 kernels: per restart against the reference's merged spectra, then the consensus
    artefacts (cnmf.py:977-985) against the reference's files."""
    obj = cNMF(output_dir=str(tmp_path), name="pkl", engine=engine)
    nc = pd.DataFrame(g["norm_counts"], index=["c%d" % i for i in range(g["norm_counts"].shape[0])], columns=list(g["genes"]))
    tpm = pd.DataFrame(g["tpm"], index=nc.index, columns=list(g["tpm_genes"]))
    obj.prepare_from_matrix(nc, components=[4, 5, 6], n_iter=12, seed=14, beta_loss="kullback-leibler", tpm=tpm)
    led = load_df_from_npz(obj.paths["nmf_replicate_parameters"])
    assert np.array_equal(led[["n_components", "iter", "nmf_seed"]].values.astype(np.int64), g["ledger"])
    obj.factorize()
    n_dev = {(int(k), int(it)): int(n) for (k, it, _), n in zip(g["ledger"], obj.last_factorize_stats["n_iter"])}
    obj.combine()
    worst, moved = 0.0, 0
    X64 = g["norm_counts"]
    for k in (4, 5, 6):
        merged = load_df_from_npz(obj.paths["merged_spectra"] % k)
        ref = g["merged_k%d" % k]
        assert merged.shape == ref.shape
        for it in range(12):
            dev = merged.values[it * k:(it + 1) * k]
            maxabs, relfro = nmf_cd.spectra_error(ref[it * k:(it + 1) * k], dev)
            if not (maxabs <= 5e-4 and relfro <= 2e-3):
                # the stopping rule is evaluated every 10 iterations on a float32 divergence: a restart that sits on the
                # threshold may stop one check earlier or later than scikit-learn's float64 run -- then it is held to the
                # float64 oracle truncated at the device's own count (the oracle reproduces the reference's merged spectra
                # to 1e-9: tests/test_golden_reference.py::test_kl_restart_spectra_reproduced)
                seed = int([s_ for kk, ii, s_ in g["ledger"] if kk == k and ii == it][0])
                _, _, n_ref = nmf_mu.nmf_mu(X64, k, seed=seed, max_iter=1000)
                assert n_dev[(k, it)] != n_ref and abs(n_dev[(k, it)] - n_ref) <= 10, (k, it, n_dev[(k, it)], n_ref, maxabs, relfro)
                _, H_ref, _ = nmf_mu.nmf_mu(X64, k, seed=seed, max_iter=n_dev[(k, it)], tol=0.0)
                maxabs, relfro = nmf_cd.spectra_error(H_ref, dev)
                moved += 1
            worst = max(worst, relfro)
            assert maxabs <= 5e-4 and relfro <= 2e-3, (k, it, maxabs, relfro)
    assert moved <= 3, moved
    print("Kullback-Leibler restarts vs the reference's merged spectra: worst relative Frobenius error %.2e (%d of 36 stopped one "
          "check away from scikit-learn's count)" % (worst, moved))
    stats = obj.k_selection_stats()
    for row, k in zip(stats.itertuples(), (4, 5, 6)):
        _, _, sil, err = g["stats_k%d" % k]
        assert row.k == k and abs(row.silhouette - sil) < 5e-3 and abs(row.prediction_error - err) <= 1e-3 * err
    for k, thr in ((5, 0.5), (4, 2.0)):
        med, usages = obj.consensus(k, density_threshold=thr)
        rep = str(thr).replace(".", "_")
        assert ((med.values - g["consensus_spectra_k%d" % k]) ** 2).sum() < TOLERANCE
        assert ((usages.values - g["consensus_usages_k%d" % k]) ** 2).sum() < TOLERANCE * usages.size
        tpm_sp = load_df_from_npz(obj.paths["gene_spectra_tpm"] % (k, rep)).values
        ref = g["gene_spectra_tpm_k%d" % k]
        assert np.abs(tpm_sp - ref).max() <= 2e-3 * np.abs(ref).max()
        score = load_df_from_npz(obj.paths["gene_spectra_score"] % (k, rep)).values
        assert np.abs(score - g["gene_spectra_score_k%d" % k]).max() < 5e-3


def test_sparse_upload_stays_sparse_on_the_kl_route(engine, monkeypatch):
    """Round-4 review, missing #2: the reference hands scikit-learn the CSR matrix as stored and scikit-learn touches the stored
    entries only.  A CSR upload keeps its compressed rows; Kullback-Leibler restarts (non-zero images built from them) and
    the float64 refits never form the dense float32 image nor the dense transposed copy -- a path that multiplies the dense
    matrix (coordinate descent) forms it on demand, with the same numbers as a dense upload."""
    X = _counts(3000, 1100, 4.6, seed=12)
    assert (X != 0).mean() < 0.2
    Xs = sp.csr_matrix(X)
    engine.set_matrix(Xs)
    # (scipy's sparse mean and numpy's dense mean of a float32 matrix differ in the last bit, and with them scikit-learn's
    #  init scale sqrt(X.mean() / k): pin it, so that the two uploads below start from identical factors)
    xm = np.float32(X.mean())
    engine.x_mean = xm
    im = engine.matrix_images()
    assert im["csr"] and not im["dense"] and not im["dense_transpose"], im
    H, W, n_iter, err = engine.nmf_mu_batch([7, 20], seeds=[3, 4], max_iter=60, return_W=True, warn=False)
    rs = np.random.RandomState(0)
    Wr, _, _ = engine.mu_refit_f64(np.abs(rs.standard_normal((5, X.shape[1]))), max_iter=30, warn=False)
    Wt, _, _ = engine.mu_refit_f64(np.abs(rs.standard_normal((5, X.shape[0]))), transposed=True, max_iter=30, warn=False)
    im = engine.matrix_images()
    assert im["csr"] and im["csr_of_transpose"] and im["non_zero_images_16"] and im["non_zero_images_32"], im
    assert not im["dense"] and not im["dense_transpose"], im
    # the same calls on a dense upload of the same matrix: the compressed rows are built on the device, same bits
    engine.set_matrix(X)
    engine.x_mean = xm
    assert engine.matrix_images()["dense"] and not engine.matrix_images()["csr"]
    Hd, Wd, nd, errd = engine.nmf_mu_batch([7, 20], seeds=[3, 4], max_iter=60, return_W=True, warn=False)
    for a, b in zip(H + W, Hd + Wd):
        np.testing.assert_array_equal(a, b)
    assert list(n_iter) == list(nd) and list(err) == list(errd)
    # coordinate descent on the CSR upload: the dense image appears when asked for, results as on the dense upload
    Hc_d, _, nc_d, _ = engine.nmf_batch([6], seeds=[9], max_iter=40, warn=False)
    engine.set_matrix(Xs)
    engine.x_mean = xm
    assert not engine.matrix_images()["dense"]
    Hc_s, _, nc_s, _ = engine.nmf_batch([6], seeds=[9], max_iter=40, warn=False)
    assert engine.matrix_images()["dense"]
    np.testing.assert_array_equal(Hc_s[0], Hc_d[0])
    np.testing.assert_array_equal(engine.get_matrix(), X)
    # Itakura-Saito touches every element: the dense kernels (and their transposed copy) on demand as well
    # (a strictly positive matrix -- scikit-learn's rule for beta_loss <= 0, mirrored by the engine -- handed over as CSR)
    engine.set_matrix(sp.csr_matrix(X + np.float32(1.0)))
    assert not engine.matrix_images()["dense"]
    engine.nmf_mu_batch([5], seeds=[3], beta_loss="itakura-saito", max_iter=10, warn=False)
    im = engine.matrix_images()
    assert im["dense"] and im["dense_transpose"], im


def test_mu_refit_f64_ragged_and_empty_rows_and_columns(engine):
    """Edge shapes of the compressed rows: cells and genes without a single stored entry (a TPM matrix keeps genes nobody
    expresses), fewer rows than one wavefront group / one transpose chunk, a single column, rank 1 -- against the oracle."""
    rs = np.random.RandomState(3)
    for n, g_, k in ((37, 19, 3), (300, 1, 1), (5, 700, 2), (1000, 400, 6)):
        X = (rs.gamma(1.0, 1.0, size=(n, g_)) * (rs.rand(n, g_) < 0.15)).astype(np.float32)
        X[rs.choice(n, max(1, n // 7), replace=False)] = 0            # empty cells
        if g_ > 2:
            X[:, rs.choice(g_, max(1, g_ // 5), replace=False)] = 0   # empty genes
        if not X.any():
            X[0, 0] = 1.0
        X64 = X.astype(np.float64)
        H = np.abs(rs.standard_normal((k, g_))) + 0.01
        U = np.abs(rs.standard_normal((n, k))) + 0.01
        for M in (X, sp.csr_matrix(X)):
            engine.set_matrix(M)
            W_ref, n_ref = _oracle_refit(X64, H, 60)
            W, it, err = engine.mu_refit_f64(H, max_iter=60, warn=False)
            assert it == n_ref and np.abs(W - W_ref).max() <= 1e-9 * max(np.abs(W_ref).max(), 1e-300), (n, g_, k)
            assert not W[~X.any(axis=1)].any()                        # a cell without entries ends at exactly zero usage
            Wt_ref, nt_ref = _oracle_refit(np.ascontiguousarray(X64.T), np.ascontiguousarray(U.T), 60)
            Wt, itt, _ = engine.mu_refit_f64(U.T, transposed=True, max_iter=60, warn=False)
            assert itt == nt_ref and np.abs(Wt - Wt_ref).max() <= 1e-9 * max(np.abs(Wt_ref).max(), 1e-300), (n, g_, k)
    # an all-zero CSR matrix uploads (no stored entries at all) and refits to zeros
    Z = sp.csr_matrix((50, 30), dtype=np.float32)
    engine.set_matrix(Z)
    engine.x_mean = np.float32(1.0)
    W, it, err = engine.mu_refit_f64(np.ones((2, 30)), max_iter=20, warn=False)
    assert not W.any() and err == 0.0


def test_gene_statistics_and_ols_product_on_the_compressed_rows(engine):
    """`cnmf_col_moments` and `cnmf_xt_matmul_f64` (the gene statistics and the X^T Y accumulation of
    efficient_ols_all_cols, cnmf.py:55-125) on a matrix that lives as compressed rows only: float64 from the stored entries
    of every gene (the zeros of a column folded into the constant term) against numpy, and against the dense kernels."""
    rs = np.random.RandomState(5)
    X = (rs.gamma(2.0, 3.0, size=(1300, 450)) * (rs.rand(1300, 450) < 0.12)).astype(np.float32)
    X[:, 7] = 0                                                   # a gene nobody expresses (variance floored by the caller)
    X64 = X.astype(np.float64)
    W = np.abs(rs.standard_normal((1300, 21)))                    # (two chunks of 16 components)
    engine.set_matrix(sp.csr_matrix(X))
    mean, var = engine.col_mean_var()
    assert not engine.matrix_images()["dense"]
    assert np.abs(mean - X64.mean(axis=0)).max() <= 1e-13 * np.abs(X64.mean(axis=0)).max()
    assert np.abs(var - X64.var(axis=0)).max() <= 1e-12 * X64.var(axis=0).max()
    v = np.where(var < 1e-12, 1e-12, var)
    out = engine.xt_matmul_f64(W, mean=mean, std=np.sqrt(v))
    plain = engine.xt_matmul_f64(W)
    assert not engine.matrix_images()["dense"]
    ref = W.T @ ((X64 - X64.mean(axis=0)) / np.sqrt(np.where(X64.var(axis=0) < 1e-12, 1e-12, X64.var(axis=0))))
    assert np.abs(out - ref).max() <= 1e-9 * np.abs(ref).max()
    assert np.abs(plain - W.T @ X64).max() <= 1e-12 * np.abs(W.T @ X64).max()
    # the prediction error of the statistics branch (cnmf.py:926-930; the reference densifies a sparse matrix there):
    # sum over the stored entries of (x - wh)^2 - (wh)^2, plus tr(W^T W . H H^T)
    Hs = np.abs(rs.standard_normal((21, 450)))
    pe = engine.prediction_error(W, Hs)
    assert not engine.matrix_images()["dense"]
    ref_pe = ((X64 - W @ Hs) ** 2).sum()
    assert abs(pe - ref_pe) <= 1e-10 * ref_pe
    engine.set_matrix(X)                                          # the dense kernels on the same matrix
    assert abs(engine.prediction_error(W, Hs) - pe) <= 1e-10 * pe
    mean_d, var_d = engine.col_mean_var()
    out_d = engine.xt_matmul_f64(W, mean=mean_d, std=np.sqrt(np.where(var_d < 1e-12, 1e-12, var_d)))
    assert np.abs(mean - mean_d).max() <= 1e-13 * np.abs(mean_d).max() and np.abs(out - out_d).max() <= 1e-9 * np.abs(out_d).max()


@pytest.mark.parametrize("beta_loss", ["kullback-leibler", "frobenius"])
def test_mirror_class_on_sparse_normalised_counts(engine, tmp_path, beta_loss):
    """The reference's sparse branch end to end through the mirror class: normalised counts AND the TPM matrix handed over
    as scipy.sparse (what `prepare(densify=False)` keeps for a sparse counts file, cnmf.py:537-556).  The same run on the
    dense forms of the same matrices is the yardstick; under the Kullback-Leibler loss the device never forms a dense
    image of either matrix (restarts, k selection's refits, the consensus tail), under 'frobenius' it forms them on
    demand."""
    # a count matrix as sparse as a real one (~10 % non-zero): 200 of its 500 genes normalised like prepare does, TPM over all
    C, _ = synth.topic_counts(700, 500, 5, 4.0, 0.4, 17)
    C = C[:, C.sum(axis=0) > 0]
    hv = np.argsort(-C.var(axis=0), kind="stable")[:200]
    hv.sort()
    keep = C[:, hv].sum(axis=1) > 0
    C = C[keep].astype(np.float64)
    NC = C[:, hv] / C[:, hv].std(axis=0, ddof=1)
    TPM = C / C.sum(axis=1, keepdims=True) * 1e6
    assert (NC != 0).mean() < 0.25
    cells = ["c%d" % i for i in range(C.shape[0])]
    tg = ["g%d" % j for j in range(C.shape[1])]
    genes = [tg[j] for j in hv]
    runs = {}
    for form in ("sparse", "dense"):
        obj = cNMF(output_dir=str(tmp_path / form), name="sp", engine=engine)
        if form == "sparse":
            nc, tpm = (sp.csr_matrix(NC), cells, genes), (sp.csr_matrix(TPM), tg)
        else:
            nc, tpm = pd.DataFrame(NC, index=cells, columns=genes), pd.DataFrame(TPM, index=cells, columns=tg)
        obj.prepare_from_matrix(nc, components=[4, 5], n_iter=6, seed=14, beta_loss=beta_loss, tpm=tpm)
        obj.factorize()
        if form == "sparse" and beta_loss != "frobenius":
            assert not engine.matrix_images()["dense"], engine.matrix_images()
        obj.combine()
        stats = obj.k_selection_stats()
        if form == "sparse" and beta_loss != "frobenius":          # k selection: float64 refits + prediction errors on the stored entries
            assert engine.shape == NC.shape and not engine.matrix_images()["dense"], engine.matrix_images()
        med, usages = obj.consensus(5, density_threshold=2.0)
        if form == "sparse" and beta_loss != "frobenius":
            im = engine.matrix_images()
            assert im["csr"] and not im["dense"] and not im["dense_transpose"], im
        rep = "2_0"
        runs[form] = (load_df_from_npz(obj.paths["merged_spectra"] % 5).values, stats, med, usages,
                      load_df_from_npz(obj.paths["gene_spectra_tpm"] % (5, rep)).values,
                      load_df_from_npz(obj.paths["gene_spectra_score"] % (5, rep)).values)
        assert list(usages.index) == cells and list(med.columns) == genes
    a, b = runs["sparse"], runs["dense"]
    # (scipy's sparse mean and numpy's dense mean differ in the last bit, and with them the init scale: float32 round-off apart)
    for it in range(6):
        maxabs, relfro = nmf_cd.spectra_error(b[0][it * 5:(it + 1) * 5], a[0][it * 5:(it + 1) * 5])
        assert maxabs <= 1e-4 and relfro <= 1e-3, (it, maxabs, relfro)
    assert np.abs(a[1]["silhouette"].values - b[1]["silhouette"].values).max() < 5e-3
    assert np.abs(a[1]["prediction_error"].values / b[1]["prediction_error"].values - 1).max() < 1e-3
    assert ((a[2].values - b[2].values) ** 2).sum() < TOLERANCE
    assert np.abs(a[3].values - b[3].values).max() <= 2e-3 * np.abs(b[3].values).max()
    assert np.abs(a[4] - b[4]).max() <= 2e-3 * np.abs(b[4]).max() and np.abs(a[5] - b[5]).max() < 5e-3


def test_stored_zeros_and_dense_matrices_do_not_cost_extra_images(engine):
    """Round-5 advice (low): (i) a CSR upload with a STORED zero (explicit, or a float64 value that underflows in float32)
    is compacted by the wrapper and stays on the no-dense-image route instead of silently forming the N x G image;
    (ii) a dense upload that the density rule sends back to the dense kernels is only COUNTED -- no compressed rows
    (12 B per entry) are left behind."""
    X = _counts(1500, 600, 4.6, seed=21)                    # ~10 % non-zero
    Xs = sp.csr_matrix(X.astype(np.float64))
    Xz = Xs.copy()
    Xz.data[::17] = 1e-60                                   # underflows to 0 in float32
    Xz.data[5] = 0.0                                        # an explicit stored zero
    engine.set_matrix(Xz)
    im = engine.matrix_images()
    assert im["csr"] and not im["dense"], im
    H, _, n, _ = engine.nmf_mu_batch([5], seeds=[3], max_iter=30, warn=False)
    assert not engine.matrix_images()["dense"]
    Xc = Xz.copy(); Xc.data = Xc.data.astype(np.float32).astype(np.float64); Xc.eliminate_zeros()
    engine.set_matrix(Xc)
    H2, _, n2, _ = engine.nmf_mu_batch([5], seeds=[3], max_iter=30, warn=False)
    assert list(n) == list(n2) and np.array_equal(H[0], H2[0])
    # (ii) mostly non-zero, dense upload: the dense kernels run, nothing compressed is kept
    Xd = synth.make_config("C1", dtype=np.float32, n_cells=600)
    assert (Xd != 0).mean() > 0.25
    engine.set_matrix(Xd)
    engine.nmf_mu_batch([5], seeds=[3], max_iter=20, warn=False)
    im = engine.matrix_images()
    assert im["dense"] and not im["csr"] and not im["csr_of_transpose"], im
