"""FULL-SIZE parity against scikit-learn's own output (float64), committed as golden vectors.

The float64 reference cannot run hundreds of outer iterations at 50 000 x 2000 inside the GPU box's time
budget, so `tools/make_golden_big.py` ran it in the build container (unmodified scikit-learn 1.7.2 through
`oracle/sklearn_ref.py`, the reference's own call, cnmf.py:672) and committed the spectra:

* `tests/golden/ref_c3_long.npz` -- C3 (the bench shape), k = 5, 11, 13: ledger seeds whose runs need 421, 760
  and 1000 outer iterations, i.e. the regime the benchmark spends its time in (k != K_true, long trajectories),
  after exactly 150 iterations and at the stopping rule;
* `tests/golden/ref_c4_csr.npz` -- C4 (200 000 x 2000 CSR), K = 20, two restarts x 10 iterations.

The device runs them on the DEFAULT path inside a full 256-column batch that needs refills and repacking.
No assertion here is conditional.
"""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from cnmf_amd import synth
from cnmf_amd.cnmf import ledger_seeds
from oracle import nmf_cd

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


_C4_CACHE = {}


def _c4_counts_csr():
    """The count-valued C4 matrix (200 000 x 2000, Poisson counts / std) as CSR, built once per test session: the
    Poisson draws cost a minute of host time."""
    if "X" not in _C4_CACHE:
        _C4_CACHE["X"] = sp.csr_matrix(synth.make_config("C4", dtype=np.float32))
    return _C4_CACHE["X"]


def _fillers(n9=20, n13=5, n7=6, seed=77):
    ks = [9] * n9 + [13] * n13 + [7] * n7
    seeds = [int(s) for s in np.random.RandomState(seed).randint(1, 2 ** 31 - 1, size=len(ks))]
    return ks, seeds


def test_C3_long_restarts_vs_sklearn_golden(engine):
    g = np.load(os.path.join(GOLD, "ref_c3_long.npz"))
    X = synth.make_config("C3", dtype=np.float32)
    X64 = X.astype(np.float64)
    assert np.allclose([X64.sum(), (X64 * X64).sum()], g["x_checksum"], rtol=1e-12)      # the very same input
    led = {(k, it): int(s) for k, it, s in ledger_seeds(list(range(5, 14)), 100, 14)}
    ks3 = [5, 11, 13]
    seeds3 = []
    for k in ks3:
        seed, it, n_full = (int(v) for v in g["k%d_seed" % k])
        assert led[(k, it)] == seed and n_full >= 300          # a real ledger row of the north-star job, and a long one
        seeds3.append(seed)
    engine.set_matrix(X)
    fk, fs = _fillers()
    ks, seeds = ks3 + fk, seeds3 + fs                          # 29 + 287 = 316 columns > 256: queue + refill
    # (a) the same truncations as the oracle: 50 and 150 outer iterations, 256 columns live, default (count) path.
    #     k != K_true = 9 trajectories are ill-conditioned -- rounding-level perturbations are amplified with the
    #     iteration count; the golden file records how far scikit-learn's OWN float32 path has drifted from its
    #     float64 path on the same restart at the same truncation (k = 13 at 150 iterations: 8e-4 / 4e-4).  The
    #     device must stay within the usual 1e-4 / 1e-3, or within 4 x that measured float32 drift where float32
    #     itself cannot do better.
    for T in (50, 150):
        H, _, n_iter, _ = engine.nmf_batch(ks, seeds=seeds, max_iter=T, warn=False, kc_max=256)
        st = engine.last_stats
        assert st["kc"] == 256 and st["gemm_mode"] >= 3
        for r, k in enumerate(ks3):
            assert int(n_iter[r]) == T
            dev = g["k%d_f32dev%d" % (k, T)]
            maxabs, relfro = nmf_cd.spectra_error(g["k%d_H%d" % (k, T)], H[r])
            assert maxabs <= max(1e-4, 4 * dev[0]) and relfro <= max(1e-3, 4 * dev[1]), (k, T, maxabs, relfro, dev)
    # (a') the same at 50 iterations in a WIDE batch (512 packed columns = two component groups per GEMM pass: stream-K
    #      pass A walking 392 tiles component-group-major on 256 workgroups, pass B spread over the XCDs by (row tile,
    #      group, K split)) -- the width large jobs run at by default
    fk2, fs2 = _fillers(seed=78)
    H, _, n_iter, _ = engine.nmf_batch(ks + fk2, seeds=seeds + fs2, max_iter=50, warn=False, kc_max=512)
    st = engine.last_stats
    assert st["kc"] == 512 and st["gemm_mode"] >= 3
    for r, k in enumerate(ks3):
        assert int(n_iter[r]) == 50
        dev = g["k%d_f32dev50" % k]
        maxabs, relfro = nmf_cd.spectra_error(g["k%d_H50" % k], H[r])
        assert maxabs <= max(1e-4, 4 * dev[0]) and relfro <= max(1e-3, 4 * dev[1]), (k, "wide", maxabs, relfro, dev)
    # (a'') THE BENCH'S OPERATING POINT (round-3 review, weak #3): the count path (gemm_mode 4) in the widest batch -- 1024
    #       packed columns = four component groups, 784 pass-A tiles walked component-group-major on 256 persistent
    #       workgroups (xmap), pass B over 8 K splits -- with MORE packed columns than the batch holds (1 251 > 1 024: queue, refill,
    #       re-packing), at the 50- and the 150-iteration truncation against the same golden spectra
    fk4, fs4 = _fillers(n9=70, n13=10, n7=25, seed=80)
    ks_w, seeds_w = ks3 + fk + fk4, seeds3 + fs + fs4
    assert sum(ks_w) >= 1040
    for T in (50, 150):
        H, _, n_iter, _ = engine.nmf_batch(ks_w, seeds=seeds_w, max_iter=T, warn=False)      # default width for this job
        st = engine.last_stats
        assert st["kc"] == 1024 and st["gemm_mode"] == 4, st
        for r, k in enumerate(ks3):
            assert int(n_iter[r]) == T
            dev = g["k%d_f32dev%d" % (k, T)]
            maxabs, relfro = nmf_cd.spectra_error(g["k%d_H%d" % (k, T)], H[r])
            assert maxabs <= max(1e-4, 4 * dev[0]) and relfro <= max(1e-3, 4 * dev[1]), (k, "1024 columns", T, maxabs, relfro, dev)
    # (b) to the stopping rule (tol 1e-4, max_iter 1000): iteration count, objective and spectra.  Calibrated like (a)
    #     (round-3 review, weak #2): the golden file records where scikit-learn's OWN float32 path lands relative to its
    #     float64 path AT THE STOPPING RULE on the same restart (tools/make_golden_big.py c3drift: k = 5: 6e-8 / 5e-7,
    #     k = 11: 4e-5 / 2e-5, k = 13 -- stops at max_iter without converging -- 7e-4 / 4e-4; the same iteration count as
    #     float64 in all three).  The device is held to the stated 1e-4 / 1e-3, or to 4 x that float32 drift where float32
    #     itself cannot do better; the iteration count to max(3, 1 %); the objective -- a stable functional -- to 2e-5.
    H, W, n_iter, viol = engine.nmf_batch(ks, seeds=seeds, warn=False, return_W=True)       # default width: 512 for 316 columns
    assert engine.last_stats["kc"] == 512 and engine.last_stats["gemm_mode"] >= 3
    for r, k in enumerate(ks3):
        n_full = int(g["k%d_seed" % k][2])
        dev = g["k%d_f32devfull" % k]
        assert int(g["k%d_f32nfull" % k][0]) == n_full                 # (scikit-learn float32 stops where float64 does)
        obj = engine.prediction_error(W[r], H[r])
        obj_ref = float(g["k%d_objfull" % k][0])
        maxabs, relfro = nmf_cd.spectra_error(g["k%d_Hfull" % k], H[r])
        print("C3 k=%d at the stopping rule: device n_iter %d (sklearn %d), spectra maxabs %.2e relfro %.2e "
              "(sklearn float32 vs float64: %.2e %.2e), objective rel. diff %.1e"
              % (k, int(n_iter[r]), n_full, maxabs, relfro, dev[0], dev[1], abs(obj - obj_ref) / obj_ref))
        assert abs(int(n_iter[r]) - n_full) <= max(3, n_full // 100), (k, int(n_iter[r]), n_full)
        assert abs(obj - obj_ref) <= 2e-5 * obj_ref, (k, obj, obj_ref)
        assert maxabs <= max(1e-4, 4 * dev[0]) and relfro <= max(1e-3, 4 * dev[1]), (k, maxabs, relfro, dev)
        if n_full < 1000:
            assert viol[r] <= 1e-4
    # (c) the GENERAL path on the same matrix (count detection off: what a Harmony-corrected or TPM-normalised input of
    #     this size gets -- X itself as two f16 planes with a per-row exponent, gemm_mode 5), at the 50-iteration truncation:
    #     a 256-column batch, and the widest one (1024 columns = four component groups of the two-plane kernels)
    engine.set_count_detection(False)
    try:
        engine.set_matrix(X)
        fk3, fs3 = _fillers(n9=100, n13=10, n7=10, seed=79)
        for width, kk, ss in ((256, ks, seeds), (1024, ks + fk3, seeds + fs3)):
            H, _, n_iter, _ = engine.nmf_batch(kk, seeds=ss, max_iter=50, warn=False, kc_max=width)
            st = engine.last_stats
            assert st["kc"] == width and st["gemm_mode"] == 5, st
            for r, k in enumerate(ks3):
                assert int(n_iter[r]) == 50
                dev = g["k%d_f32dev50" % k]
                maxabs, relfro = nmf_cd.spectra_error(g["k%d_H50" % k], H[r])
                assert maxabs <= max(1e-4, 4 * dev[0]) and relfro <= max(1e-3, 4 * dev[1]), (k, "general", width, maxabs, relfro, dev)
    finally:
        engine.set_count_detection(True)


def test_C3_nndsvd_init_vs_sklearn_golden(engine):
    """init='nndsvd' at the FULL bench size (50 000 x 2000; the other NNDSVD tests stop at 9 000 cells): the device range
    finder (Cholesky-QR over 64 chunks of 784 cells, 2 x 7 + 2 passes over X) + the host tail against scikit-learn's
    _initialize_nmf in float64 (tools/make_golden_big.py c3nndsvd): H0 in full, W0 on every 97th cell and through its
    column sums / sums of squares over all cells."""
    g = np.load(os.path.join(GOLD, "ref_c3_nndsvd.npz"))
    X = synth.make_config("C3", dtype=np.float32)
    X64 = X.astype(np.float64)
    assert np.allclose([X64.sum(), (X64 * X64).sum()], g["x_checksum"], rtol=1e-12)
    del X64
    engine.set_matrix(X)
    ks, seeds = [9, 13], [int(g["k9_seed"][0]), int(g["k13_seed"][0])]
    for (k, seed), (W0, H0) in zip(zip(ks, seeds), engine.nndsvd_init_batch(ks, seeds)):
        H_ref, W_rows, W_sums = g["k%d_H0" % k], g["k%d_W0_rows" % k], g["k%d_W0_colsum" % k]
        assert W0.shape == (X.shape[0], k) and H0.shape == H_ref.shape
        eh = np.abs(H0 - H_ref).max() / np.abs(H_ref).max()
        ew = np.abs(W0[::97] - W_rows).max() / np.abs(W_rows).max()
        es = np.abs(np.array([W0.sum(axis=0), (W0 * W0).sum(axis=0)]) - W_sums).max(axis=1) / np.abs(W_sums).max(axis=1)
        print("C3 nndsvd k=%d: H0 %.2e, W0 rows %.2e, W0 column sums %.2e / %.2e of the largest entry" % (k, eh, ew, es[0], es[1]))
        assert eh <= 1e-3 and ew <= 1e-3 and (es <= 1e-3).all(), (k, eh, ew, es)


def _c4_matrix():
    rs = np.random.RandomState(3)
    X = sp.random(200_000, 2000, density=0.08, format="csr", dtype=np.float32, random_state=rs,
                  data_rvs=lambda n: rs.gamma(1.0, 1.0, size=n).astype(np.float32))
    return X[np.asarray(X.sum(axis=1)).ravel() > 0]


def test_C4_csr_restarts_vs_sklearn_golden(engine):
    g = np.load(os.path.join(GOLD, "ref_c4_csr.npz"))
    X = _c4_matrix()
    assert tuple(g["shape"]) == X.shape
    assert abs(float(X.data.astype(np.float64).sum()) - float(g["x_checksum"][0])) <= 1e-9 * float(g["x_checksum"][0])
    engine.set_matrix(X)                                       # CSR upload, densified on the device
    # two restarts alone (exact-f32 pipe) ...
    H, _, n_iter, _ = engine.nmf_batch([20, 20], seeds=[11, 12], max_iter=10, warn=False)
    for r, seed in enumerate((11, 12)):
        assert int(n_iter[r]) == 10
        maxabs, relfro = nmf_cd.spectra_error(g["seed%d_H10" % seed], H[r])
        assert maxabs <= 1e-4 and relfro <= 1e-3, (seed, maxabs, relfro)
    # ... and inside a full-width batch (13 x 20 = 260 columns: 256-column split-operand kernels + one refill)
    ks = [20] * 13
    seeds = [11, 12] + list(range(201, 212))
    H, _, n_iter, _ = engine.nmf_batch(ks, seeds=seeds, max_iter=10, warn=False, kc_max=256)
    assert engine.last_stats["kc"] == 256 and engine.last_stats["gemm_mode"] >= 2
    for r, seed in enumerate((11, 12)):
        assert int(n_iter[r]) == 10
        maxabs, relfro = nmf_cd.spectra_error(g["seed%d_H10" % seed], H[r])
        assert maxabs <= 1e-4 and relfro <= 1e-3, (seed, maxabs, relfro)


def test_C4_count_valued_csr_vs_sklearn_golden(engine):
    """BASELINE config 4 as it occurs in practice: a COUNT-valued 200 000 x 2000 matrix (Poisson counts / std, handed
    over as CSR like a sparse h5ad) -- the default f16 integer-plane kernels (gemm_mode 4) with 782 cell tiles, i.e.
    ~3 tiles per persistent stream-K workgroup, a shape no smaller test reaches.  scikit-learn's float64 output after
    10 iterations (tools/make_golden_big.py c4counts) vs the device inside a 260-column batch: 1e-4 / 1e-3."""
    g = np.load(os.path.join(GOLD, "ref_c4_counts.npz"))
    X = _c4_counts_csr()
    assert tuple(g["shape"]) == X.shape and int(g["nnz"][0]) == X.nnz
    assert abs(float(X.data.astype(np.float64).sum()) - float(g["x_checksum"][0])) <= 1e-9 * float(g["x_checksum"][0])
    engine.set_matrix(X)                                       # CSR upload, densified on the device
    # 13 x 20 = 260 columns: a 256-column batch with one refill, then the default width for that job (512);
    # 52 x 20 = 1040 columns: the widest batch (1024 columns = four component groups over 782 cell tiles) with one refill
    for kc, n_rest, width in ((256, 13, 256), (0, 13, 512), (0, 52, 1024)):
        ks = [20] * n_rest
        seeds = [21, 22] + list(range(301, 299 + n_rest))
        H, _, n_iter, _ = engine.nmf_batch(ks, seeds=seeds, max_iter=10, warn=False, kc_max=kc)
        st = engine.last_stats
        assert st["kc"] == width and st["gemm_mode"] == 4, st             # the count structure was detected: f16 two-plane path
        for r, seed in enumerate((21, 22)):
            assert int(n_iter[r]) == 10
            maxabs, relfro = nmf_cd.spectra_error(g["seed%d_H10" % seed], H[r])
            assert maxabs <= 1e-4 and relfro <= 1e-3, (kc, seed, maxabs, relfro)


def test_C4_as_BASELINE_states_it_to_the_stopping_rule(engine):
    """BASELINE config 4 AS WRITTEN: the 200 000-cell CSR matrix, K = 20, the n_iter = 100 ledger (seed 14) run to
    scikit-learn's stopping rule in ONE device call at the default width (1024 packed columns = 51 restarts in flight over
    782 cell tiles, refill, the tail narrowing) -- round-4 review, weak #3: C4 was never run past 10 iterations.  On this
    matrix K = 20 = K_true: every restart stops after 36..55 outer iterations; scikit-learn's float64 output for the
    shortest and the longest of them (tools/make_golden_big.py c4stop; its own float32 path stops at the same counts,
    1e-7 / 1e-6 away) vs the device: iteration count exact, spectra 1e-4 / 1e-3, objective 1e-5."""
    g = np.load(os.path.join(GOLD, "ref_c4_stop.npz"))
    X = _c4_counts_csr()
    assert tuple(g["shape"]) == X.shape
    assert abs(float(X.data.astype(np.float64).sum()) - float(g["x_checksum"][0])) <= 1e-9 * float(g["x_checksum"][0])
    led = ledger_seeds([20], 100, 14)
    engine.set_matrix(X)
    H, _, n_iter, _ = engine.nmf_batch([k for k, _, _ in led], seeds=[int(s) for _, _, s in led], warn=False)
    st = engine.last_stats
    assert st["kc"] == 1024 and st["gemm_mode"] == 4, st
    assert 30 <= int(n_iter.min()) and int(n_iter.max()) <= 70            # (the whole ledger stops where scikit-learn's rows do)
    for row in (15, 75):
        seed, it, n_ref, n_ref32 = (int(v) for v in g["row%d_seed" % row])
        assert int(led[row][2]) == seed and n_ref == n_ref32
        assert int(n_iter[row]) == n_ref, (row, int(n_iter[row]), n_ref)
        maxabs, relfro = nmf_cd.spectra_error(g["row%d_H" % row], H[row])
        print("C4 K=20 ledger row %d: %d iterations (scikit-learn %d), spectra %.2e / %.2e" % (row, n_iter[row], n_ref, maxabs, relfro))
        assert maxabs <= 1e-4 and relfro <= 1e-3, (row, maxabs, relfro)
    # the objective of the two restarts (their usages come back too: a second call, the two rows in front of 50 others)
    rows = [15, 75] + [r for r in range(52) if r not in (15, 75)][:50]
    H2, W2, n2, _ = engine.nmf_batch([20] * 52, seeds=[int(led[r][2]) for r in rows], warn=False, return_W=True)
    assert engine.last_stats["kc"] == 1024
    for i, row in enumerate((15, 75)):
        assert int(n2[i]) == int(g["row%d_seed" % row][2])
        obj = engine.prediction_error(W2[i], H2[i])
        ref = float(g["row%d_obj" % row][0])
        assert abs(obj - ref) <= 1e-5 * ref, (row, obj, ref)


def _c4kl_csr():
    """The matrix of tools/make_golden_big.py::c4kl_matrix: C4's topic model at a real 10x library size (~9 % non-zero)."""
    if "KL" not in _C4_CACHE:
        C, _ = synth.topic_counts(200_000, 2000, 20, 5.2, 0.4, 3)
        _C4_CACHE["KL"] = sp.csr_matrix(synth.normalise_like_prepare(C, dtype=np.float32))
    return _C4_CACHE["KL"]


def test_C4_kl_non_zero_path_vs_sklearn_on_csr_golden(engine, monkeypatch):
    """Round-4 review, weak #1 / next #1a: the Kullback-Leibler non-zero kernels at the size they were benchmarked at --
    200 000 x 2000, 36 M stored entries (98 blocks of cells: their float32 partial numerators summed in block order) --
    against scikit-learn ITSELF on the CSR matrix (`non_negative_factorization(X_csr, solver='mu',
    beta_loss='kullback-leibler')`, float64, tools/make_golden_big.py c4kl): K = 20 and K = 9 after 20 and 100
    iterations.  The path is taken by the density rule (asserted: no dense image is ever formed, and the forced path gives
    the same bits); spectra 1e-4 / 1e-3 or 4 x scikit-learn's own float32-vs-float64 distance where that is larger, the
    divergence 1e-3, usages 2e-3; bit identity across batch compositions at this size."""
    g = np.load(os.path.join(GOLD, "ref_c4_kl.npz"))
    X = _c4kl_csr()
    assert tuple(g["shape"]) == X.shape and int(g["nnz"][0]) == X.nnz
    assert abs(float(X.data.astype(np.float64).sum()) - float(g["x_checksum"][0])) <= 1e-9 * float(g["x_checksum"][0])
    assert X.nnz < 0.25 * X.shape[0] * X.shape[1]
    engine.set_matrix(X)
    cases = [(k, T) for k in (20, 9) for T in (20, 100) if "k%d_H%d" % (k, T) in g.files]
    assert len(cases) == 4, cases
    singles = {}
    for k, T in cases:
        seed = int(g["k%d_seed" % k][0])
        H, W, n_iter, err = engine.nmf_mu_batch([k], seeds=[seed], max_iter=T, return_W=True, warn=False)
        im = engine.matrix_images()
        assert im["non_zero_images_%d" % (16 if k <= 16 else 32)] and not im["dense"] and not im["dense_transpose"], im
        n_ref, n_ref32 = (int(v) for v in g["k%d_n%d" % (k, T)])
        assert int(n_iter[0]) == n_ref, (k, T, int(n_iter[0]), n_ref, n_ref32)
        maxabs, relfro = nmf_cd.spectra_error(g["k%d_H%d" % (k, T)], H[0])
        d32 = g["k%d_f32dev%d" % (k, T)]
        bar = (max(1e-4, 4 * float(np.nan_to_num(d32[0]))), max(1e-3, 4 * float(np.nan_to_num(d32[1]))))
        ref_err = float(g["k%d_err%d" % (k, T)][0])
        print("C4 KL k=%d T=%d: %d iterations, spectra %.2e / %.2e (scikit-learn float32: %.2e / %.2e), divergence %.9g vs %.9g"
              % (k, T, n_iter[0], maxabs, relfro, d32[0], d32[1], err[0], ref_err))
        assert maxabs <= bar[0] and relfro <= bar[1], (k, T, maxabs, relfro, bar)
        assert abs(err[0] - ref_err) <= 1e-3 * ref_err
        Wh = g["k%d_Whead%d" % (k, T)]
        # (same seed, same initial factors: the components come in the same order)
        assert np.abs(W[0][:4096] - Wh).max() <= 2e-3 * np.abs(Wh).max()
        assert np.abs(W[0].sum(axis=0) - g["k%d_Wsum%d" % (k, T)]).max() <= 1e-3 * np.abs(g["k%d_Wsum%d" % (k, T)]).max()
        singles[(k, T)] = (H[0], W[0], int(n_iter[0]), float(err[0]))
    # forced == chosen by density; another batch composition (both padded ranks in flight, fillers around) == alone
    monkeypatch.setenv("CNMF_MU_SPARSE", "1")
    Hf, Wf, nf, ef = engine.nmf_mu_batch([20], seeds=[int(g["k20_seed"][0])], max_iter=20, return_W=True, warn=False)
    monkeypatch.delenv("CNMF_MU_SPARSE")
    np.testing.assert_array_equal(Hf[0], singles[(20, 20)][0]); np.testing.assert_array_equal(Wf[0], singles[(20, 20)][1])
    ks = [13, 20, 5, 9, 20, 9]
    seeds = [7, int(g["k20_seed"][0]), 8, int(g["k9_seed"][0]), 9, 10]
    Hb, Wb, nb, eb = engine.nmf_mu_batch(ks, seeds=seeds, max_iter=20, return_W=True, warn=False)
    np.testing.assert_array_equal(Hb[1], singles[(20, 20)][0]); np.testing.assert_array_equal(Wb[1], singles[(20, 20)][1])
    if (9, 20) in singles:
        np.testing.assert_array_equal(Hb[3], singles[(9, 20)][0]); np.testing.assert_array_equal(Wb[3], singles[(9, 20)][1])
    assert float(eb[1]) == singles[(20, 20)][3]


def test_C3_pipeline_consensus_vs_sklearn_f64_golden(engine):
    """Round-5 review, next #1 -- what users consume is the CONSENSUS, and the reference's only value-pinning test is on
    consensus outputs (/root/reference/tests/test_reproducibility.py:12, 96-115).  The whole pipeline at the HEADLINE shape:
    the C3 matrix (50 000 x 2000), the reference's ledger for components 7 / 9 / 11 with n_iter = 8 and seed 14; golden =
    every restart by scikit-learn in FLOAT64 to the stopping rule (422 ... 1000 iterations at K = 11: the regime the bench
    spends its time in), merged, through the consensus core and the stats branch of oracle/consensus.py
    (tools/make_golden_big.py c3pipe -> tests/golden/ref_c3_pipeline.npz).  Device: the same ledger rows on the DEFAULT
    path at the bench's operating point -- the f16 count kernels, 1024 packed columns with queue and refill (filler restarts
    from the north-star ledger make the batch as wide as the bench's; asserted) -- then the device's merged spectra through
    the device's consensus.  Bars: consensus spectra at the reference's own sum((a - b)^2) < 1e-4, UNCONDITIONALLY, for all
    three K; usages, silhouette and prediction error are asserted at the tightest bar they were MEASURED to meet and printed
    next to what scikit-learn's own float32 pipeline moves them by (recorded in the golden file)."""
    g = np.load(os.path.join(GOLD, "ref_c3_pipeline.npz"))
    X = synth.make_config("C3", dtype=np.float32)
    X64 = X.astype(np.float64)
    assert tuple(g["shape"]) == X.shape and np.allclose([X64.sum(), (X64 * X64).sum()], g["x_checksum"], rtol=1e-12)
    Ks = [int(k) for k in g["ks"]]
    n_per_k = int(g["n_iter_per_k"][0])
    led = ledger_seeds(Ks, n_per_k, 14)
    assert np.array_equal(np.array([(k, it, s) for k, it, s in led], dtype=np.int64), g["ledger"])
    # fillers: the first 12 ledger rows of every K of the north-star job (972 columns) -> 1 188 packed columns in the queue
    north = ledger_seeds(list(range(5, 14)), 100, 14)
    fill = [(k, int(s)) for k, it, s in north if it < 12]
    ks = [k for k, _, _ in led] + [k for k, _ in fill]
    seeds = [int(s) for _, _, s in led] + [s for _, s in fill]
    engine.set_matrix(X)
    H, _, n_iter, _ = engine.nmf_batch(ks, seeds=seeds, warn=False)
    st = engine.last_stats
    assert st["kc"] == 1024 and st["gemm_mode"] == 4, st                    # the bench's operating point
    H, n_iter = H[:len(led)], n_iter[:len(led)]
    report = {}
    for K in Ks:
        rows = [i for i, (k, _, _) in enumerate(led) if k == K]
        n_dev, n_ref = n_iter[rows].astype(np.int64), g["k%d_n_iter" % K].astype(np.int64)
        merged = np.concatenate([H[i] for i in rows], axis=0).astype(np.float64)       # (iter asc, topic asc): cnmf.py:765-770
        out = engine.consensus(merged, K, density_threshold=0.5)
        ref = g["k%d_median_spectra" % K]
        med = out["median_spectra"]
        perm, cos = nmf_cd.match_components(ref, med)                               # cluster ids may be permuted
        assert sorted(perm) == list(range(K)), (K, perm)
        med = med[perm]
        sumsq = float(((med - ref) ** 2).sum())
        rel = float(np.linalg.norm(med - ref) / np.linalg.norm(ref))
        kept_dev, kept_ref = int(out["n_kept"]), int(g["k%d_density_filter" % K].sum())
        dens = float(np.abs(out["local_density"] - g["k%d_local_density" % K]).max())
        # usages of the consensus spectra (cnmf.py:919) on the float64 device refit
        W, _ = engine.nnls_f64(out["median_spectra"][perm], warn=False)
        cs_ref = g["k%d_usage_colsum" % K]
        u_rel = float(np.abs(W.sum(axis=0) - cs_ref[0]).max() / cs_ref[0].max())
        u_rows = float(np.abs(W[::97] - g["k%d_usage_rows" % K]).max() / np.abs(g["k%d_usage_rows" % K]).max())
        # the stats branch (cnmf.py:922-936): no density filter, silhouette + prediction error
        sout = engine.consensus(merged, K, skip_density=True, want_silhouette=True)
        Ws, _ = engine.nnls_f64(sout["median_spectra"], warn=False)
        perr = engine.prediction_error(Ws, sout["median_spectra"])
        sil_d = abs(sout["silhouette"] - float(g["k%d_silhouette" % K][0]))
        perr_rel = abs(perr / float(g["k%d_prediction_error" % K][0]) - 1.0)
        drift = g["k%d_f32_drift" % K] if ("k%d_f32_drift" % K) in g.files else None
        report[K] = dict(n_iter_device=n_dev.tolist(), n_iter_sklearn_f64=n_ref.tolist(), consensus_sumsq=sumsq, consensus_rel=rel,
                         min_cos=float(cos.min()), kept=(kept_dev, kept_ref), max_density_diff=dens, usage_colsum_rel=u_rel,
                         usage_rows_rel=u_rows, silhouette_absdiff=sil_d, prediction_error_rel=perr_rel,
                         sklearn_f32_pipeline_drift=(None if drift is None else [float("%.3g" % v) for v in drift]))
        print("C3 pipeline K=%d: %s" % (K, report[K]))
    for K in Ks:
        r = report[K]
        # the reference's TOLERANCE on the artefact users consume -- unconditional, for every K
        assert r["consensus_sumsq"] < 1e-4, (K, r)
        assert r["consensus_rel"] <= 1e-3 and r["min_cos"] > 0.9999, (K, r)
        assert r["kept"][0] == r["kept"][1], (K, r)                                 # the density filter keeps the same spectra
        assert r["max_density_diff"] <= 1e-3, (K, r)
        assert r["usage_colsum_rel"] <= 1e-3 and r["usage_rows_rel"] <= 2e-3, (K, r)
        # k selection's two statistics (cnmf.py:922-936): scikit-learn's OWN float32 pipeline moves the silhouette by
        # 4e-7 / 5e-9 / 2.1e-6 and the prediction error by 2e-10 / 2e-13 / 4e-8 relative at K = 7 / 9 / 11 (golden file)
        assert r["prediction_error_rel"] <= 1e-6, (K, r)
        assert r["silhouette_absdiff"] <= 1e-5, (K, r)
