"""BASELINE.json configurations at (or near) full size on the GPU.

C2  2700 x 2000 (PBMC3k stand-in), K=10, n_iter=100: the whole ledger through the slot
    work-queue; a sample of restarts is compared with the float64 oracle.
C4  200 000 x 2000 sparse (CSR, ~8 % dense), K=20: densify-on-device + size-independent
    properties (checksum of the densified matrix, determinism, non-negativity, monotone
    objective) -- the CPU oracle cannot finish this size in seconds.
"""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from cnmf_amd import synth
from cnmf_amd.cnmf import ledger_seeds
from oracle import nmf_cd, sklearn_ref

pytestmark = pytest.mark.gpu


def test_C2_full_ledger_sampled_parity(engine):
    X = synth.make_config("C2", dtype=np.float64)
    assert X.shape[0] == 2700 and X.shape[1] <= 2000
    engine.set_matrix(X)
    led = ledger_seeds([10], 100, 14)                      # the product's own ledger
    ks = [k for k, _, _ in led]
    seeds = [s for _, _, s in led]
    H, _, n_iter, viol = engine.nmf_batch(ks, seeds=seeds, warn=False)
    # round 6: a job of >= 512 columns runs wide on a small matrix too (1 000 columns in ONE 1024-column batch on the f16
    # count kernels instead of four 256-column rounds: 17.5 -> 8.6 ms per job)
    assert engine.last_stats["kc"] == 1024 and engine.last_stats["gemm_mode"] == 4, engine.last_stats
    assert len(H) == 100 and all(h.shape == (10, X.shape[1]) for h in H)
    assert (n_iter >= 1).all() and (n_iter <= 1000).all()
    conv = n_iter < 1000
    assert (viol[conv] <= 1e-4).all()                       # sklearn's stopping rule held on the device
    assert all(np.isfinite(h).all() and (h >= 0).all() for h in H)
    st = engine.last_stats
    assert st["restart_iterations"] == int(n_iter.sum())
    # sample: the five fastest-converging restarts against the float64 oracle
    for r in np.argsort(n_iter)[:5]:
        _, H_ref, n_ref = nmf_cd.nmf(X, 10, seed=seeds[r])
        maxabs, relfro = nmf_cd.spectra_error(H_ref, H[r])
        assert abs(int(n_iter[r]) - n_ref) <= max(3, n_ref // 100), (r, n_iter[r], n_ref)
        assert maxabs <= (1e-4 if n_ref <= 500 else 5e-4) and relfro <= 1e-3, (r, maxabs, relfro)


def test_C4_sparse_densify_on_device_properties(engine):
    rs = np.random.RandomState(3)
    N, G, k = 200_000, 2000, 20
    X = sp.random(N, G, density=0.08, format="csr", dtype=np.float32, random_state=rs,
                  data_rvs=lambda n: rs.gamma(1.0, 1.0, size=n).astype(np.float32))
    X = X[np.asarray(X.sum(axis=1)).ravel() > 0]
    engine.set_matrix(X)
    N = X.shape[0]
    # (1) checksum of the densified matrix: ||X - 0||^2 on the device == sum of squares of the CSR data
    zW = np.zeros((N, 1)); zH = np.zeros((1, G))
    ss = engine.prediction_error(zW, zH)
    ref = float((X.data.astype(np.float64) ** 2).sum())
    assert abs(ss - ref) <= 1e-6 * ref
    # (2) two K=20 restarts, 25 outer iterations: deterministic, non-negative, finite
    H1, W1, n1, _ = engine.nmf_batch([k, k], seeds=[11, 12], max_iter=25, return_W=True, warn=False)
    H2, W2, n2, _ = engine.nmf_batch([k, k], seeds=[11, 12], max_iter=25, return_W=True, warn=False)
    assert list(n1) == [25, 25] and list(n2) == [25, 25]
    for a, b in zip(H1 + W1, H2 + W2):
        assert np.array_equal(a, b)                          # no atomics / fixed reduction orders
        assert np.isfinite(a).all() and (a >= 0).all()
    # (3) the objective after 25 iterations is far below the objective after 2
    H3, W3, _, _ = engine.nmf_batch([k], seeds=[11], max_iter=2, return_W=True, warn=False)
    e25 = engine.prediction_error(W1[0], H1[0])
    e2 = engine.prediction_error(W3[0], H3[0])
    assert e25 < e2 < ss
    # (4) NNLS refit at full size reproduces the usages of a converged W half-step:
    #     refitting with H fixed can only lower the objective
    Wr, _ = engine.nnls(H1[0], max_iter=200)
    assert engine.prediction_error(Wr, H1[0]) <= e25 * (1 + 1e-6)
    # (5) 256 packed columns at this size (1563 cell tiles: ~6 per persistent workgroup of the stream-K
    #     launch): the split-operand path against the exact-f32 matrix pipe, and bit-reproducible
    ks12, seeds12 = [k] * 12, list(range(101, 113))
    Ha, _, na, _ = engine.nmf_batch(ks12, seeds=seeds12, max_iter=12, warn=False)
    assert engine.last_stats["kc"] == 256 and engine.last_stats["gemm_mode"] == 5      # gamma-valued: the general f16 path
    Hb, _, nb, _ = engine.nmf_batch(ks12, seeds=seeds12, max_iter=12, warn=False)
    assert all(np.array_equal(a, b) for a, b in zip(Ha, Hb))
    import os
    os.environ["CNMF_GEMM3"] = "0"
    try:
        Hc, _, _, _ = engine.nmf_batch(ks12, seeds=seeds12, max_iter=12, warn=False)
        assert engine.last_stats["gemm_mode"] == 0
    finally:
        del os.environ["CNMF_GEMM3"]
    for a, c in zip(Ha, Hc):
        assert np.isfinite(a).all() and np.abs(a - c).max() <= 1e-4 * max(1.0, np.abs(c).max())


def test_C3_full_size_restarts_vs_sklearn(engine):
    """The north-star shape itself (50 000 x 2000): restarts at the data's true rank converge in
    ~40 iterations, so scikit-learn (float64) can serve as the oracle at full size."""
    X = synth.make_config("C3", dtype=np.float32)
    assert X.shape == (50000, 2000)
    engine.set_matrix(X)
    ks, seeds = [9, 9, 5], [817790314, 59886188, 1812018521]
    H, _, n_iter, viol = engine.nmf_batch(ks, seeds=seeds, warn=False)
    X64 = X.astype(np.float64)
    for r in (0, 1):
        H_ref, _, n_ref = sklearn_ref.nmf(X64, ks[r], seeds[r])
        maxabs, relfro = nmf_cd.spectra_error(H_ref, H[r])
        assert abs(int(n_iter[r]) - n_ref) <= 3, (n_iter[r], n_ref)
        assert maxabs <= 1e-4 and relfro <= 1e-3, (maxabs, relfro)
    assert (viol[n_iter < 1000] <= 1e-4).all()
    # size-independent property at full size: re-running is bit-identical (fixed reduction orders)
    H2, _, n2, _ = engine.nmf_batch(ks, seeds=seeds, warn=False)
    assert list(n2) == list(n_iter) and all(np.array_equal(a, b) for a, b in zip(H, H2))


def test_streamk_split_operand_path_at_scale(engine, monkeypatch):
    """26 000 cells x 2000 genes with 256 packed columns: pass A runs as the stream-K launch of the
    split-operand GEMM (tiles cut once or twice between persistent workgroups, partial planes added by
    the sweep), pass B as its split-K launch.  25 outer iterations of 29 rank-9 restarts: three of them
    against scikit-learn (float64) stopped at the same iteration, all of them against the exact-f32
    matrix pipe, and twice for bit-reproducibility."""
    X = synth.make_config("C3", dtype=np.float32, n_cells=26000)
    engine.set_matrix(X)
    ks = [9] * 29
    seeds = [int(s) for s in np.random.RandomState(5).randint(1, 2**31 - 1, size=29)]
    monkeypatch.setenv("CNMF_GEMM3", "2")
    H, _, n_iter, _ = engine.nmf_batch(ks, seeds=seeds, max_iter=25, warn=False, kc_max=256)
    st = engine.last_stats
    assert st["kc"] == 256 and st["gemm_mode"] == 2
    assert list(n_iter) == [25] * 29
    H2, _, _, _ = engine.nmf_batch(ks, seeds=seeds, max_iter=25, warn=False, kc_max=256)
    assert all(np.array_equal(a, b) for a, b in zip(H, H2))
    X64 = X.astype(np.float64)
    for r in (0, 14, 28):
        H_ref, _, n_ref = sklearn_ref.nmf(X64, 9, seeds[r], max_iter=25)
        assert n_ref == 25
        maxabs, relfro = nmf_cd.spectra_error(H_ref, H[r])
        assert maxabs <= 1e-4 and relfro <= 1e-3, (r, maxabs, relfro)
    monkeypatch.setenv("CNMF_GEMM3", "0")
    H0, _, n0, _ = engine.nmf_batch(ks, seeds=seeds, max_iter=25, warn=False)
    assert engine.last_stats["gemm_mode"] == 0
    for a, b in zip(H, H0):
        assert np.abs(a - b).max() <= 1e-4 * max(1.0, np.abs(b).max())
    # the default: count structure detected (the synthetic matrix is counts / std) -> integer-plane kernels,
    # here with few 256-cell tiles (102 < 192: K split + reduce instead of stream-K)
    refs = {r: sklearn_ref.nmf(X64, 9, seeds[r], max_iter=25)[0] for r in (0, 14, 28)}
    for mode in ("3", "4"):                               # three bf16 planes / two f16 planes (the default)
        monkeypatch.setenv("CNMF_GEMM3", mode)
        Hc, _, nc, _ = engine.nmf_batch(ks, seeds=seeds, max_iter=25, warn=False)
        assert engine.last_stats["gemm_mode"] == int(mode)
        Hc2, _, _, _ = engine.nmf_batch(ks, seeds=seeds, max_iter=25, warn=False)
        assert all(np.array_equal(a, b) for a, b in zip(Hc, Hc2))
        for r in (0, 14, 28):
            maxabs, relfro = nmf_cd.spectra_error(refs[r], Hc[r])
            assert maxabs <= 1e-4 and relfro <= 1e-3, (mode, r, maxabs, relfro)


def test_C3_full_width_default_path_soak(engine):
    """The north-star shape with 256 packed columns on the default path (count structure detected ->
    stream-K launch of the integer-plane kernel over 196 cell tiles, cut pieces added by the sweep):
    three runs must agree bit for bit (the kernels synchronise by hand-counted vmcnt / raw barriers),
    and two restarts are checked against scikit-learn stopped at the same iteration."""
    X = synth.make_config("C3", dtype=np.float32)
    engine.set_matrix(X)
    ks = [9] * 20 + [13] * 5 + [5] * 2                      # 20*9 + 65 + 10 = 255 columns
    seeds = [int(s) for s in np.random.RandomState(9).randint(1, 2**31 - 1, size=len(ks))]
    runs = []
    for _ in range(3):
        H, _, n_iter, _ = engine.nmf_batch(ks, seeds=seeds, max_iter=30, warn=False)
        assert engine.last_stats["kc"] == 256 and engine.last_stats["gemm_mode"] == 4
        runs.append(H)
    for other in runs[1:]:
        assert all(np.array_equal(a, b) for a, b in zip(runs[0], other))
    X64 = X.astype(np.float64)
    for r in (0, 22):
        H_ref, _, n_ref = sklearn_ref.nmf(X64, ks[r], seeds[r], max_iter=30)
        maxabs, relfro = nmf_cd.spectra_error(H_ref, runs[0][r])
        assert maxabs <= 1e-4 and relfro <= 1e-3, (r, maxabs, relfro)


def test_mid_size_job_is_promoted_to_the_count_kernels(engine):
    """65..128 packed columns of a large count-structured matrix run on the 256-column integer-plane kernels
    (half empty) rather than on 128 columns of the f32 pipe; the result is the usual one."""
    X = synth.make_config("C3", dtype=np.float32, n_cells=10000)
    engine.set_matrix(X)
    ks, seeds = [9] * 12, list(range(21, 33))               # 108 columns
    H, _, n_iter, _ = engine.nmf_batch(ks, seeds=seeds, max_iter=25, warn=False)
    assert engine.last_stats["kc"] == 256 and engine.last_stats["gemm_mode"] == 4
    H_ref, _, _ = sklearn_ref.nmf(X.astype(np.float64), 9, seeds[3], max_iter=25)
    maxabs, relfro = nmf_cd.spectra_error(H_ref, H[3])
    assert maxabs <= 1e-4 and relfro <= 1e-3, (maxabs, relfro)
    Hk, _, _, _ = engine.nmf_batch(ks, seeds=seeds, max_iter=25, warn=False, kc_max=128)   # explicit cap: f32 pipe
    assert engine.last_stats["kc"] == 128 and engine.last_stats["gemm_mode"] == 0
    for a, b in zip(H, Hk):
        assert np.abs(a - b).max() <= 1e-4 * max(1.0, np.abs(b).max())


def test_reduce_folded_into_the_H_sweep_is_bit_identical(engine, monkeypatch):
    """On the f16 count path the split-K partial planes of pass B are summed (split order) and scaled by the per-gene
    constant INSIDE the H half-step; the separate reduce kernel (CNMF_NO_PSUM=1) must give the same bits."""
    X = synth.make_config("C3", dtype=np.float32, n_cells=12000)
    engine.set_matrix(X)
    ks = [9] * 20 + [13] * 4 + [5] * 4
    seeds = list(range(301, 301 + len(ks)))
    H, _, n_iter, _ = engine.nmf_batch(ks, seeds=seeds, max_iter=40, warn=False)
    assert engine.last_stats["kc"] == 256 and engine.last_stats["gemm_mode"] == 4
    monkeypatch.setenv("CNMF_NO_PSUM", "1")
    H2, _, n2, _ = engine.nmf_batch(ks, seeds=seeds, max_iter=40, warn=False)
    assert list(n2) == list(n_iter) and all(np.array_equal(a, b) for a, b in zip(H, H2))


@pytest.mark.parametrize("K", [10, 13])
def test_C2_full_pipeline_consensus_spectra_vs_cpu_reference(engine, K):
    """What users consume is the CONSENSUS spectra, not single restarts: BASELINE config 2 end to end -- the full
    ledger (n_iter = 100, seed 14, restarts run to the stopping rule) on the device, the device's merged spectra
    through the device's consensus (density filter 0.5, KMeans, medians) -- against the same pipeline computed ENTIRELY
    on the CPU reference path (scikit-learn float64 restarts + the consensus core; tools/make_golden_c2.py ->
    tests/golden/ref_c2_consensus.npz).  K = 10 is the config as written (= the data's own rank: ~35 iterations per
    restart); K = 13 on the same matrix gives the long, ill-conditioned trajectories the bench spends its time in
    (159 ... 1000 iterations, mean 464).  Bars: the reference's own sum((a - b)^2) < 1e-4
    (tests/test_reproducibility.py:12) and -- because that is lax for rows that sum to 1 over 2000 genes -- 1e-3
    relative (scikit-learn's own float32 pipeline sits 6e-6 / 4e-5 from its float64 one at K = 13: recorded in the
    golden file as calibration)."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_c2_consensus.npz"))
    X = synth.make_config("C2", dtype=np.float64)
    assert tuple(g["shape"]) == X.shape and np.allclose([X.sum(), (X * X).sum()], g["x_checksum"], rtol=1e-12)
    engine.set_matrix(X)
    led = ledger_seeds([K], 100, 14)
    H, _, n_iter, _ = engine.nmf_batch([k for k, _, _ in led], seeds=[int(s) for _, _, s in led], warn=False)
    # iteration counts of the device vs scikit-learn float64, restart by restart (same ledger order)
    n_ref = g["k%d_n_iter" % K].astype(np.int64)
    close = np.abs(n_iter.astype(np.int64) - n_ref) <= np.maximum(3, n_ref // 20)
    assert close.mean() >= 0.9, (n_iter[~close], n_ref[~close])
    merged = np.concatenate(H, axis=0).astype(np.float64)               # (iter asc, topic asc) like combine_nmf
    out = engine.consensus(merged, K, density_threshold=0.5)
    assert abs(int(out["n_kept"]) - int(g["k%d_n_kept" % K][0])) <= 13  # <= 1 % of the spectra sit at the threshold
    med, ref = out["median_spectra"], g["k%d_median_spectra" % K]
    perm, cos = nmf_cd.match_components(ref, med)                       # cluster ids may be permuted
    assert sorted(perm) == list(range(K)) and cos.min() > 0.9999
    med = med[perm]
    assert ((med - ref) ** 2).sum() < 1e-4                              # the reference's TOLERANCE
    assert np.linalg.norm(med - ref) <= 1e-3 * np.linalg.norm(ref)
    assert np.abs(med - ref).max() <= 1e-3 * np.abs(ref).max()


@pytest.mark.parametrize("wide", ["1", "0"])
def test_small_matrix_wide_batch_matches_oracle(engine, monkeypatch, wide):
    """Round 6 (`wide_small` in batch_host.hip.h): the tutorial-sized matrix (1000 x 500 -> 1024 x 512 padded: four cell tiles,
    two gene tiles) with a job of 120 restarts (~840 columns): the batch is as wide as the job (CNMF_WIDE_SMALL=0: 256 columns,
    the round-5 rule), every sampled restart against its float64 oracle run, and the two widths agree restart by restart."""
    monkeypatch.setenv("CNMF_WIDE_SMALL", wide)
    X = synth.make_config("C1", dtype=np.float64)
    engine.set_matrix(X)
    rs = np.random.RandomState(23)
    ks = [int(k) for k in rs.randint(5, 10, size=120)]
    seeds = [int(s) for s in rs.randint(1, 2**31 - 1, size=120)]
    H, _, n_iter, viol = engine.nmf_batch(ks, seeds=seeds, warn=False)
    st = engine.last_stats
    assert st["kc"] == (1024 if wide == "1" else 256) and st["gemm_mode"] == 4, st
    assert (viol[n_iter < 1000] <= 1e-4).all()
    for r in range(0, 120, 9):
        _, H_ref, n_ref = nmf_cd.nmf(X, ks[r], seed=seeds[r])
        maxabs, relfro = nmf_cd.spectra_error(H_ref, H[r])
        assert abs(int(n_iter[r]) - n_ref) <= max(3, n_ref // 100), (r, n_iter[r], n_ref)
        assert maxabs <= 1e-4 and relfro <= 1e-3, (r, maxabs, relfro)
