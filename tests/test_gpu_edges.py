"""Edge cases of the restart engine through the C-ABI: tiny / ragged shapes (below every
tile size), rank 1 and rank CNMF_KMAX, an empty restart list, all-zero rows, and
regularisation strong enough to zero a factor (violation_init == 0 path)."""
import numpy as np
import pytest

from cnmf_amd import synth
from oracle import nmf_cd

pytestmark = pytest.mark.gpu


def _x(n, g, seed=0):
    rs = np.random.RandomState(seed)
    return np.abs(rs.standard_normal((n, g))) * (rs.rand(n, g) < 0.6) + 0.01


@pytest.mark.parametrize("n,g,k", [(7, 5, 2), (33, 31, 3), (129, 33, 5), (257, 65, 1), (200, 140, 32), (64, 128, 9),
                                   (300, 150, 33), (260, 200, 48), (400, 180, 64),
                                   (300, 170, 65), (500, 260, 96), (1100, 300, 128)])
def test_small_and_ragged_shapes(engine, n, g, k):
    X = _x(n, g, seed=n + g)
    engine.set_matrix(X)
    W_ref, H_ref, n_ref = nmf_cd.nmf(X, k, seed=5, max_iter=300)
    H, W, n_iter, _ = engine.nmf_batch([k], seeds=[5], max_iter=300, return_W=True, warn=False)
    assert H[0].shape == (k, g) and W[0].shape == (n, k)
    assert abs(int(n_iter[0]) - n_ref) <= max(3, n_ref // 50)
    if int(n_iter[0]) != n_ref:       # a few iterations apart: compare at the device's own truncation
        W_ref, H_ref, _ = nmf_cd.nmf(X, k, seed=5, max_iter=int(n_iter[0]), tol=0.0)
    # same init, same component order: compare the reconstructions (robust to near-degenerate factors)
    R, R_ref = W[0].astype(np.float64) @ H[0], W_ref @ H_ref
    assert np.abs(R - R_ref).max() <= 2e-3 * max(1.0, np.abs(R_ref).max())


def test_ranks_above_64_next_to_small_ones_and_their_refits(engine):
    """Ranks 65 .. 128 (sweep_big_kernel: w in registers, p / running w in per-lane LDS strips, Gram as its own launch)
    in ONE batch with small ranks (the register-resident tiers), to sklearn's stopping rule; then the NNLS refit and the
    prediction error at rank 100, and the rank limit itself."""
    X = _x(700, 260, seed=21)
    engine.set_matrix(X)
    ks = [5, 128, 9, 70, 33, 96, 64, 65]
    seeds = list(range(31, 31 + len(ks)))
    H, W, n_iter, viol = engine.nmf_batch(ks, seeds=seeds, max_iter=80, return_W=True, warn=False)
    for k, s_, h, w, n in zip(ks, seeds, H, W, n_iter):
        W_ref, H_ref, n_ref = nmf_cd.nmf(X, k, seed=s_, max_iter=80)
        assert h.shape == (k, 260) and w.shape == (700, k) and abs(int(n) - n_ref) <= 2, (k, int(n), n_ref)
        if int(n) != n_ref:
            W_ref, H_ref, _ = nmf_cd.nmf(X, k, seed=s_, max_iter=int(n), tol=0.0)
        R, R_ref = w.astype(np.float64) @ h, W_ref @ H_ref
        assert np.abs(R - R_ref).max() <= 2e-3 * max(1.0, np.abs(R_ref).max()), k
    H2, _, n2, _ = engine.nmf_batch(ks, seeds=seeds, max_iter=80, warn=False)
    assert list(n2) == list(n_iter) and all(np.array_equal(a, b) for a, b in zip(H, H2))      # deterministic
    # big and small ranks side by side on the split-operand kernels (a matrix they take: count-structured C1), in a 256-column
    # batch and three times over in the widest one (1024 packed columns, four component groups): same answers to rounding
    from cnmf_amd import synth
    Xc = synth.make_config("C1", dtype=np.float64)
    engine.set_matrix(Xc)
    Hn, _, nn, _ = engine.nmf_batch(ks, seeds=seeds, max_iter=60, warn=False, kc_max=256)
    assert engine.last_stats["gemm_mode"] == 4, engine.last_stats
    Hw, _, nw, _ = engine.nmf_batch(ks * 3, seeds=seeds * 3, max_iter=60, warn=False, kc_max=1024)
    assert engine.last_stats["kc"] == 1024 and engine.last_stats["gemm_mode"] == 4, engine.last_stats
    for i in range(3 * len(ks)):
        a, na = Hn[i % len(ks)], int(nn[i % len(ks)])
        assert abs(int(nw[i]) - na) <= 2, (ks[i % len(ks)], int(nw[i]), na)
        if int(nw[i]) == na:
            maxabs, relfro = nmf_cd.spectra_error(a, Hw[i])
            assert maxabs <= 1e-4 and relfro <= 1e-3, (ks[i % len(ks)], maxabs, relfro)
    _, Hc_ref, n_ref = nmf_cd.nmf(Xc, 96, seed=seeds[5], max_iter=60)
    assert abs(int(nn[5]) - n_ref) <= 2
    if int(nn[5]) == n_ref:
        maxabs, relfro = nmf_cd.spectra_error(Hc_ref, Hn[5])
        assert maxabs <= 1e-4 and relfro <= 1e-3, (maxabs, relfro)
    engine.set_matrix(X)
    _, Hr, _ = nmf_cd.nmf(X, 100, seed=3, max_iter=30)
    W_ref, n_ref = nmf_cd.nnls(X, Hr, max_iter=60)
    Wd, nd = engine.nnls(Hr, max_iter=60, warn=False)
    assert abs(nd - n_ref) <= 2 and np.abs(Wd - W_ref).max() <= 2e-3 * np.abs(W_ref).max()
    ref = ((X - W_ref @ Hr) ** 2).sum()
    assert abs(engine.prediction_error(W_ref, Hr) - ref) <= 1e-6 * ref
    with pytest.raises(NotImplementedError):
        engine.nmf_batch([129], seeds=[1], max_iter=5)


def test_consensus_with_more_than_64_clusters(engine):
    """KMeans / medians / silhouette of the consensus core with k = 100 clusters against the oracle."""
    from cnmf_amd import synth
    from oracle import consensus as oc
    S, _ = synth.consensus_stress(R=1200, G=150, k=100, n_outliers=40, seed=6)
    ref = oc.consensus_core(S, np.abs(np.random.RandomState(0).standard_normal((20, 150))), 100, density_threshold=0.5)
    out = engine.consensus(S, 100, density_threshold=0.5, want_silhouette=True)
    assert np.array_equal(out["density_filter"], ref["density_filter"])
    assert np.array_equal(out["labels"][out["density_filter"]] + 1, ref["kmeans_labels"])
    assert np.abs(out["median_spectra"] - ref["median_spectra"]).max() < 1e-12
    assert abs(out["silhouette"] - oc.silhouette_score(ref["l2_spectra"], ref["kmeans_labels"])) < 1e-9


def test_empty_restart_list(engine):
    engine.set_matrix(_x(20, 10))
    H, W, n_iter, viol = engine.nmf_batch([], seeds=[])
    assert H == [] and len(n_iter) == 0 and len(viol) == 0


def test_mixed_ranks_one_call(engine):
    X = _x(150, 70, seed=3)
    engine.set_matrix(X)
    ks = [1, 32, 2, 17, 16, 5, 31, 8, 40, 64, 33]
    seeds = list(range(11, 11 + len(ks)))
    H, _, n_iter, _ = engine.nmf_batch(ks, seeds=seeds, max_iter=60, warn=False)
    for k, s, h, n in zip(ks, seeds, H, n_iter):
        _, H_ref, n_ref = nmf_cd.nmf(X, k, seed=s, max_iter=60)
        assert h.shape == (k, 70) and abs(int(n) - n_ref) <= 2
        assert np.isfinite(h).all() and (h >= 0).all()


def test_zero_rows_and_columns(engine):
    X = _x(90, 40, seed=9)
    X[10:15] = 0          # cells without counts
    X[:, 7] = 0           # a gene without counts
    engine.set_matrix(X)
    W_ref, H_ref, n_ref = nmf_cd.nmf(X, 4, seed=2, max_iter=200)
    H, W, n_iter, _ = engine.nmf_batch([4], seeds=[2], max_iter=200, return_W=True, warn=False)
    assert abs(int(n_iter[0]) - n_ref) <= 3
    assert np.abs(W[0][10:15]).max() <= 1e-6 and np.abs(H[0][:, 7]).max() <= 1e-6


def test_heavy_l1_zeroes_usages(engine):
    """l1 so strong that the first sweep zeroes W: then WtW = 0, the H sweep is skipped
    (hess == 0, _cdnmf_fast.pyx:34) and sklearn stops on the violation ratio."""
    X = _x(60, 30, seed=4)
    engine.set_matrix(X)
    W_ref, H_ref, n_ref = nmf_cd.nmf(X, 3, seed=8, alpha_W=50.0, alpha_H=50.0, l1_ratio=1.0, max_iter=50)
    H, W, n_iter, _ = engine.nmf_batch([3], seeds=[8], alpha_W=50.0, alpha_H=50.0, l1_ratio=1.0, max_iter=50,
                                       return_W=True, warn=False)
    assert np.abs(W_ref).max() == 0 and np.abs(W[0]).max() == 0
    assert np.abs(H[0] - H_ref).max() <= 1e-5 * np.abs(H_ref).max()
    assert int(n_iter[0]) == n_ref


def test_set_matrix_twice_and_shapes(engine):
    engine.set_matrix(_x(50, 20))
    engine.nmf_batch([3], seeds=[1], max_iter=5, warn=False)
    engine.set_matrix(_x(31, 77))
    H, _, _, _ = engine.nmf_batch([4], seeds=[1], max_iter=5, warn=False)
    assert H[0].shape == (4, 77)
    with pytest.raises(ValueError):
        engine.nnls(np.ones((3, 20)))                       # H for the old gene count


def test_large_rank_refit_and_mu(engine):
    """Ranks above 32 through the NNLS refit and the KL multiplicative-update solver."""
    from oracle import nmf_mu
    X = _x(350, 160, seed=12)
    engine.set_matrix(X)
    _, H, _ = nmf_cd.nmf(X, 40, seed=3, max_iter=40)
    W_ref, n_ref = nmf_cd.nnls(X, H, max_iter=100)
    W, n = engine.nnls(H, max_iter=100, warn=False)
    assert abs(n - n_ref) <= 2 and np.abs(W - W_ref).max() <= 2e-3 * np.abs(W_ref).max()
    W_ref, H_ref, n_ref = nmf_mu.nmf_mu(X, 36, seed=4, max_iter=50)
    Hm, Wm, nm, _ = engine.nmf_mu_batch([36], seeds=[4], max_iter=50, return_W=True, warn=False)
    assert int(nm[0]) == n_ref
    R, R_ref = Wm[0].astype(np.float64) @ Hm[0], W_ref @ H_ref
    assert np.abs(R - R_ref).max() <= 2e-3 * np.abs(R_ref).max()
    # Itakura-Saito above rank 32 (matrix-pipe kernels at padded rank 64, as Kullback-Leibler)
    Xp = X + 0.05                                              # IS needs strictly positive data
    engine.set_matrix(Xp)
    for k_is in (40, 64):
        W_ref, H_ref, n_ref = nmf_mu.nmf_mu(Xp, k_is, seed=6, beta_loss="itakura-saito", max_iter=40)
        Hi, Wi, ni, _ = engine.nmf_mu_batch([k_is], seeds=[6], beta_loss="itakura-saito", max_iter=40, return_W=True, warn=False)
        assert abs(int(ni[0]) - n_ref) <= 10
        if int(ni[0]) != n_ref:
            W_ref, H_ref, _ = nmf_mu.nmf_mu(Xp, k_is, seed=6, beta_loss="itakura-saito", max_iter=int(ni[0]), tol=0.0)
        R, R_ref = Wi[0].astype(np.float64) @ Hi[0], W_ref @ H_ref
        assert np.abs(R - R_ref).max() <= 5e-3 * np.abs(R_ref).max(), k_is
    with pytest.raises(NotImplementedError):
        engine.nmf_mu_batch([65], seeds=[1], max_iter=5)      # CNMF_MU_KMAX = 64


@pytest.mark.parametrize("n,g,k", [(7, 5, 2), (33, 31, 3), (129, 33, 5), (257, 65, 1), (200, 140, 32), (64, 128, 16),
                                   (300, 150, 17), (131, 257, 9), (300, 150, 33), (161, 257, 48), (257, 129, 64)])
def test_kl_small_and_ragged_shapes(engine, n, g, k):
    """The Kullback-Leibler solver's matrix-pipe kernels work on 128-wide blocks and 32-deep steps: shapes below and
    across every one of those sizes, ranks 1 / 16 / 17 / 32 / 33 / 48 / 64 (all three register layouts: two restarts
    per workgroup and two M tiles above rank 32), against the float64 oracle."""
    from oracle import nmf_mu
    X = _x(n, g, seed=n + g)
    engine.set_matrix(X)
    W_ref, H_ref, n_ref = nmf_mu.nmf_mu(X, k, seed=5, max_iter=60)
    H, W, n_iter, err = engine.nmf_mu_batch([k], seeds=[5], max_iter=60, return_W=True, warn=False)
    assert H[0].shape == (k, g) and W[0].shape == (n, k)
    assert abs(int(n_iter[0]) - n_ref) <= 10
    if int(n_iter[0]) != n_ref:
        W_ref, H_ref, _ = nmf_mu.nmf_mu(X, k, seed=5, max_iter=int(n_iter[0]), tol=0.0)
    R, R_ref = W[0].astype(np.float64) @ H[0], W_ref @ H_ref
    assert np.abs(R - R_ref).max() <= 2e-3 * max(1.0, np.abs(R_ref).max())
    ref_err = nmf_mu.beta_divergence(X, W_ref, H_ref, 1, square_root=True)
    assert abs(err[0] - ref_err) <= 2e-3 * max(ref_err, 1e-6)


def test_kl_refit_regularised_and_zero_rows(engine):
    """update_H = False (the refit of cnmf.py:776-802 under beta_loss='kullback-leibler'), l1 / l2 penalties, and a
    matrix with all-zero cells and genes (denominators of zero: sklearn's EPSILON / 1.0 substitutions)."""
    from oracle import nmf_mu
    X = _x(180, 90, seed=3)
    X[[0, 17, 179]] = 0.0
    X[:, [2, 89]] = 0.0
    engine.set_matrix(X)
    W_ref, H_ref, n_ref = nmf_mu.nmf_mu(X, 6, seed=8, max_iter=40, alpha_W=0.002, alpha_H=0.001, l1_ratio=0.3)
    H, W, n_iter, _ = engine.nmf_mu_batch([6], seeds=[8], max_iter=40, alpha_W=0.002, alpha_H=0.001, l1_ratio=0.3,
                                          return_W=True, warn=False)
    assert abs(int(n_iter[0]) - n_ref) <= 10
    if int(n_iter[0]) != n_ref:
        W_ref, H_ref, _ = nmf_mu.nmf_mu(X, 6, seed=8, max_iter=int(n_iter[0]), tol=0.0, alpha_W=0.002, alpha_H=0.001, l1_ratio=0.3)
    assert np.isfinite(H[0]).all() and np.isfinite(W[0]).all()
    R, R_ref = W[0].astype(np.float64) @ H[0], W_ref @ H_ref
    assert np.abs(R - R_ref).max() <= 2e-3 * max(1.0, np.abs(R_ref).max())
    Hn = H_ref / np.maximum(H_ref.sum(axis=1, keepdims=True), 1e-12)
    Wr_ref, nr_ref = nmf_mu.nnls_mu(X, Hn, max_iter=80)
    Wr, nr = engine.nnls_mu(Hn, max_iter=80, warn=False)
    assert abs(nr - nr_ref) <= 10
    if nr != nr_ref:
        Wr_ref, _ = nmf_mu.nnls_mu(X, Hn, max_iter=int(nr), tol=0.0)
    assert np.abs(Wr - Wr_ref).max() <= 2e-3 * max(1e-6, np.abs(Wr_ref).max())


def test_round4_entry_points_at_their_edges(engine):
    """The round-4 additions of the C ABI at their edges: the float64 refit at rank 100 (the 128-accumulator product
    kernel) and on a ragged shape; pairwise distances / silhouette with a row count that is not a multiple of the 64-row
    tiles and with two clusters; queue hints outside the rank range; the spectra store under a wrong gene count; the text
    formatter's capacity check."""
    import ctypes as C
    X = _x(333, 170, seed=4)
    engine.set_matrix(X)
    rs = np.random.RandomState(9)
    # float64 refit, rank 100 and rank 3
    for k in (100, 3):
        Hs = np.abs(rs.standard_normal((k, 170)))
        W_ref, n_ref = nmf_cd.nnls(X.astype(np.float32).astype(np.float64), Hs, max_iter=200)
        W, n = engine.nnls_f64(Hs, max_iter=200, warn=False)
        assert abs(n - n_ref) <= 1 and np.abs(W - W_ref).max() <= 1e-8 * max(1.0, np.abs(W_ref).max()), k
    # distances / silhouette: 101 rows, two clusters
    from sklearn.metrics import silhouette_score
    from sklearn.metrics.pairwise import euclidean_distances
    rows = rs.rand(101, 37)
    labels = (rs.rand(101) < 0.3).astype(int)
    D, sil = engine.pairwise_distances(rows, labels=labels)
    assert np.abs(D - euclidean_distances(rows)).max() < 1e-7 and abs(sil - silhouette_score(rows, labels)) < 1e-9
    with pytest.raises(ValueError):
        engine.pairwise_distances(rows, labels=np.zeros(101, dtype=int))              # one label only
    # queue hints: ranks outside 1..CNMF_KMAX are refused, None clears
    with pytest.raises(ValueError):
        engine.set_iteration_hints({0: 10.0})
    with pytest.raises(ValueError):
        engine.set_iteration_hints({129: 10.0})
    engine.set_iteration_hints({5: 40.0, 9: 900.0})
    H1, _, n1, _ = engine.nmf_batch([5, 9, 5, 9], seeds=[1, 2, 3, 4], max_iter=30, warn=False)
    engine.set_iteration_hints(None)
    assert all(np.isfinite(h).all() for h in H1) and len(n1) == 4
    # the spectra store carries its own gene count
    engine.spectra_reset()
    first = engine.spectra_append(np.abs(rs.standard_normal((12, 50))).astype(np.float32))
    assert first == 0 and engine.spectra_rows == 12 and engine.spectra_genes == 50
    with pytest.raises(RuntimeError):
        engine.spectra_append(np.zeros((3, 51), dtype=np.float32))                      # another gene count
    with pytest.raises(ValueError):
        engine.consensus(None, 3, store_rows=[0, 1, 2, 3, 4, 99])                       # row outside the store
    out = engine.consensus(None, 3, store_rows=np.arange(12), skip_density=True)
    ref = engine.consensus(engine.spectra_fetch().astype(np.float64), 3, skip_density=True)
    assert np.array_equal(out["labels"], ref["labels"]) and np.array_equal(out["median_spectra"], ref["median_spectra"])
    engine.spectra_reset()
    # text formatter: too small a buffer reports the capacity it needs
    lib = engine._lib
    v = np.ascontiguousarray(rs.rand(4, 3))
    small = np.empty(8, dtype=np.uint8)
    need = lib.cnmf_format_rows_f64(v.ctypes.data_as(C.POINTER(C.c_double)), 4, 3, b"\t", None, 0, small.ctypes.data_as(C.c_void_p), small.size)
    assert need == -(4 * 3 * 33)
    big = np.empty(-need, dtype=np.uint8)
    n = lib.cnmf_format_rows_f64(v.ctypes.data_as(C.POINTER(C.c_double)), 4, 3, b"\t", None, 0, big.ctypes.data_as(C.c_void_p), big.size)
    assert bytes(big[:n]).decode() == "".join("\t".join(repr(float(x)) for x in row) + "\n" for row in v)


def test_matrix_beyond_2_31_padded_elements_keeps_the_matrix_pipe_path(engine):
    """Maximum sizes (round 6): 1 100 000 cells x 2 000 genes = 2.25e9 padded elements, count planes of 4.5 GB -- beyond the
    65 535 x 16 cells the plane builders' launch grid allowed until round 6 (such a matrix silently fell back to the
    exact-f32 pipe at 256 columns).  The default f16 count path at 1024 packed columns against the SAME restarts on the
    exact-f32 matrix pipe at 32 columns (other kernels, other index arithmetic): a 32-bit overflow in either shows as a
    mismatch or a fault; the LAST cells take part through a usage refit checked in float64.  (tools/probe_big_matrix.py is
    the same with independently drawn cells.)"""
    avail = [int(l.split()[1]) for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0] / 1e6
    if avail < 48:
        pytest.skip("needs ~30 GB of host memory (%.0f GB available)" % avail)
    base = synth.make_config("C3", dtype=np.float32, n_cells=100_000)
    n_tiles, n_cells = 11, 1_100_000
    X = np.empty((n_cells, base.shape[1]), dtype=np.float32)
    for t in range(n_tiles):                                  # distinct row order per tile: a wrong-tile read is a wrong row
        X[t * 100_000:(t + 1) * 100_000] = np.roll(base, 7919 * t, axis=0)
    del base
    engine.set_matrix(X)
    ks = [9] * 100 + [13] * 10                                # 1 030 columns: a 1024-wide batch with a queue
    seeds = list(range(7, 7 + len(ks)))
    H, _, n, _ = engine.nmf_batch(ks, seeds=seeds, max_iter=5, tol=0.0, warn=False)
    st = engine.last_stats
    assert st["kc"] == 1024 and st["gemm_mode"] == 4, st
    sel = [0, 57, 105]
    H32, _, _, _ = engine.nmf_batch([ks[i] for i in sel], seeds=[seeds[i] for i in sel], max_iter=5, tol=0.0, warn=False, kc_max=32)
    assert engine.last_stats["gemm_mode"] == 0
    for j, i in enumerate(sel):
        maxabs, relfro = nmf_cd.spectra_error(H32[j].astype(np.float64), H[i])
        assert maxabs <= 1e-4 and relfro <= 1e-3, (i, maxabs, relfro)
    Hn = H[0] / H[0].sum(axis=1, keepdims=True)
    Wd, _ = engine.nnls(Hn, max_iter=30, warn=False)
    tail = slice(n_cells - 2000, n_cells)
    W_ref, _ = nmf_cd.nnls(X[tail].astype(np.float64), Hn.astype(np.float64), max_iter=30)
    assert np.abs(Wd[tail] - W_ref).max() <= 2e-3 * np.abs(W_ref).max()
    engine.set_matrix(np.ones((4, 4), dtype=np.float32))      # release the 9 GB image and its planes for the tests that follow
