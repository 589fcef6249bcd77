"""GPU parity tests of the restart hot loop (run with ``pytest -m gpu`` on an MI355X).

The HIP path (through the C-ABI) is compared with the numpy restatement of sklearn's
CD solver (oracle/nmf_cd.py, pinned to sklearn in tests/test_oracle_nmf.py) and with
scikit-learn itself (the arithmetic the reference calls at cnmf.py:672) on identical
seeds / identical W0,H0.

Stated tolerance: per-restart spectra, rows L2-normalised and matched by
best cosine: max-abs <= 1e-4 and relative Frobenius <= 1e-3 versus the float64 oracle for
restarts that stop within 500 outer iterations; ill-conditioned restarts that need more
(rank far above the data's true rank -> a flat objective; fp32 round-off accumulates over
~1000 sweeps) are held to max-abs <= 5e-4 at the same relative-Frobenius bound.
|delta n_iter| is bounded too (the stop is a ratio of order-dependent fp sums).
"""
import numpy as np
import pytest

from cnmf_amd import synth
from oracle import nmf_cd

pytestmark = pytest.mark.gpu

TOL_MAXABS = 1e-4
TOL_MAXABS_LONG = 5e-4      # restarts with > 500 outer iterations
TOL_RELFRO = 1e-3


def _check(H_ref, n_ref, H, n, slack=2):
    maxabs, relfro = nmf_cd.spectra_error(H_ref, H)
    tol = TOL_MAXABS if n_ref <= 500 else TOL_MAXABS_LONG
    assert maxabs <= tol and relfro <= TOL_RELFRO, (maxabs, relfro, n_ref, n)
    assert abs(int(n) - int(n_ref)) <= max(slack, n_ref // 100), (n_ref, n)
    return maxabs, relfro


@pytest.mark.parametrize("variant", [1, 2, 3])
@pytest.mark.parametrize("mode", [0, 1])
def test_gemm_vs_numpy(engine, variant, mode):
    """The MFMA GEMM (all tile shapes) against float64 numpy; asymmetric operands."""
    rs = np.random.RandomState(5)
    for KC, K, J, ns in [(32, 64, 96, 1), (64, 224, 160, 3), (128, 2016, 512, 2), (256, 512, 2016, 4)]:
        A = rs.standard_normal((KC, K)).astype(np.float32)
        B = rs.standard_normal((J, K) if mode == 0 else (K, J)).astype(np.float32)
        ref = A.astype(np.float64) @ (B.T if mode == 0 else B).astype(np.float64)
        out, _ = engine.debug_gemm(mode, A, B, variant=variant, nsplit=ns if mode == 1 else 1)
        err = np.abs(out - ref).max() / np.abs(ref).max()
        assert err < 2e-6, (KC, K, J, ns, err)


@pytest.mark.parametrize("g3mode", ["1", "2"])
def test_split_operand_gemm_is_f32_accurate(engine, monkeypatch, g3mode):
    """The 3 x bf16 split-operand MFMA product (kernels_gemm3.hip.h) is held to the SAME bound as the
    exact-f32 matrix pipe above, on wide-dynamic-range and on non-negative sparse operands, for
    both kernel variants, with and without a K split."""
    monkeypatch.setenv("CNMF_GEMM3", g3mode)
    rs = np.random.RandomState(7)
    for K, J, ns in [(16, 40, 1), (64, 192, 1), (2016, 992, 1), (4096, 320, 4), (2048, 130, 7)]:
        A = (rs.standard_normal((256, K)) * np.exp(rs.standard_normal((256, K)))).astype(np.float32)
        B = (rs.standard_normal((J, K)) * np.exp(rs.standard_normal((J, K)))).astype(np.float32)
        ref = A.astype(np.float64) @ B.astype(np.float64).T
        out, _ = engine.debug_gemm3(A, B, nsplit=ns)
        assert np.abs(out - ref).max() / np.abs(ref).max() < 2e-6, (K, J, ns)
        # error relative to sum |a||b| (the natural scale of an inner product): f32-class
        scale = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64).T
        assert (np.abs(out - ref) / scale).max() < 3e-6, (K, J, ns)
    A = np.abs(rs.standard_normal((256, 2016))).astype(np.float32)
    B = (np.abs(rs.standard_normal((640, 2016))) * (rs.rand(640, 2016) < 0.1)).astype(np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64).T
    out, _ = engine.debug_gemm3(A, B)
    assert (np.abs(out - ref) / np.maximum(ref, 1e-30)).max() < 5e-6
    # bit-reproducible
    out2, _ = engine.debug_gemm3(A, B)
    assert np.array_equal(out, out2)


@pytest.mark.parametrize("big", [False, True])
def test_count_path_gemm_is_exact_product_accurate(engine, big):
    """Count-structured operand (kernels_counts.hip.h): B holds integers <= 256 as ONE bf16 plane (big: a few
    entries up to 65 535 -> a second, block-flagged plane), A three planes -- every partial product is exact,
    so the only error is the f32 accumulation."""
    rs = np.random.RandomState(3)
    for K, J, ns in [(16, 40, 1), (64, 300, 1), (2048, 1000, 1), (4096, 520, 4), (2048, 130, 7), (4144, 300, 1)]:
        A = (rs.standard_normal((256, K)) * np.exp(rs.standard_normal((256, K)))).astype(np.float32)
        B = rs.poisson(3.0, size=(J, K)).astype(np.float32)
        B[0, :3] = [256, 255, 0]
        if big:
            idx = rs.randint(0, J * K, size=max(3, J * K // 5000))
            B.ravel()[idx] = rs.choice([257, 300, 511, 512, 4097, 65535, 65280], size=idx.size)
        ref = A.astype(np.float64) @ B.astype(np.float64).T
        out, _ = engine.debug_gemm3c(A, B, nsplit=ns)
        assert np.abs(out - ref).max() / np.abs(ref).max() < 1e-6, (K, J, ns)
        scale = np.abs(A).astype(np.float64) @ B.astype(np.float64).T
        # (a 65 535 next to Poisson(3) entries: every later f32 accumulation rounds at the size of that one
        #  term -- the same for any f32 accumulator -- hence the wider bound for `big`)
        assert (np.abs(out - ref) / np.maximum(scale, 1e-30)).max() < (2e-5 if big else 1e-6), (K, J, ns)
        out2, _ = engine.debug_gemm3c(A, B, nsplit=ns)
        assert np.array_equal(out, out2)


@pytest.mark.parametrize("big", [False, True])
@pytest.mark.parametrize("nsub", [1, 2])
def test_f16_count_path_gemm_accuracy(engine, big, nsub):
    """The default product for count-structured data (kernels_gemm2h.hip.h): B <= 2048 as ONE f16 plane (big: a few
    entries up to 65 535 -> the flagged second plane), A >= 0 as TWO f16 planes with a per-row exponent.  The factor
    is represented to within 1 ulp_f32 (exactly for 3 values in 4), every partial product is exact: held to the
    same bounds as the three-plane bf16 product above, with rows whose scale differs by 2^+-40."""
    rs = np.random.RandomState(13)
    for K, J, ns in [(64, 40, 1), (128, 300, 1), (2048, 1000, 1), (4096, 520, 4), (2048, 130, 7), (4160, 300, 3)]:
        A = np.abs(rs.standard_normal((256, K)) * np.exp(rs.standard_normal((256, K)))).astype(np.float32)
        A[rs.rand(256, K) < 0.3] = 0.0                           # NMF factors are sparse
        A *= np.exp2(rs.randint(-40, 41, size=(256, 1))).astype(np.float32)       # per-row exponents
        A[5] = 0.0                                                # an empty row
        B = rs.poisson(3.0, size=(J, K)).astype(np.float32)
        B[0, :4] = [2048, 2047, 257, 0]
        if big:
            idx = rs.randint(0, J * K, size=max(3, J * K // 5000))
            B.ravel()[idx] = rs.choice([2049, 3000, 4097, 65535, 63488], size=idx.size)
        A64, B64 = A.astype(np.float64), B.astype(np.float64)
        ref = A64 @ B64.T
        out, _ = engine.debug_gemm2h(A, B, nsplit=ns, nsub=nsub)
        rowmax = np.abs(ref).max(axis=1, keepdims=True)
        rowmax[rowmax == 0] = 1.0
        # the bound of the exact-f32 pipe (2e-6), here per output ROW (rows differ by 2^80 in scale)
        assert (np.abs(out - ref) / rowmax).max() < 2e-6, (K, J, ns)
        assert not out[5].any()
        scale = np.maximum(A64 @ B64.T, 1e-300)
        assert (np.abs(out - ref) / scale).max() < (2e-5 if big else 3e-6), (K, J, ns)
        out2, _ = engine.debug_gemm2h(A, B, nsplit=ns, nsub=nsub)
        assert np.array_equal(out, out2)


@pytest.mark.parametrize("outlier", [1.0, 1e4, 1e6])
def test_f16_scale_bound_of_the_W_half_step_with_outlier_cells(engine, outlier):
    """Pass B (X^T.W) scales every component row of W by the BOUND the W half-step reports -- sqrt(sum w^2) over the
    <= 1024 cells of a sweep workgroup, up to 32 x the true maximum -- and ONE huge cell (a doublet, a cell with an
    enormous library: 10^4 x and 10^6 x the typical usage of its component) pushes every other entry of that row far
    below the row scale, where the two f16 planes keep an ABSOLUTE accuracy (2^-39 of the scale) rather than a
    relative one.  The product is held per element against float64, in units of |W|^T.|n| (the natural scale of the
    sum): the measured worst case is the number DESIGN.md section 4 quotes."""
    rs = np.random.RandomState(21)
    K, J = 4096, 520                                      # K = cells (the reduction), J = genes
    A = np.abs(rs.standard_normal((256, K)) * np.exp(0.5 * rs.standard_normal((256, K)))).astype(np.float32)
    A[rs.rand(256, K) < 0.3] = 0.0
    if outlier > 1.0:
        for c in range(0, 256, 2):                        # every other component has one outlier cell
            A[c, rs.randint(K)] = np.float32(outlier * (1.0 + rs.rand()))
    A[7] = 1.0                                            # a dense, flat component: the bound is its full 32 x above the maximum
    B = rs.poisson(0.4, size=(J, K)).astype(np.float32)   # sparse counts: most genes see NOTHING of the outlier cell
    A64, B64 = A.astype(np.float64), B.astype(np.float64)
    ref = A64 @ B64.T
    scale = np.maximum(np.abs(A64) @ np.abs(B64).T, 1e-300)
    worst = {}
    for bound in (False, True):
        out, _ = engine.debug_gemm2h(A, B, nsplit=4, nsub=2, sweep_bound=bound)
        worst[bound] = float((np.abs(out - ref) / scale).max())
    print("f16 scale bound: outlier %g -> max |err| / (|W|^T.|n|): exact row maximum %.3g, sweep bound %.3g"
          % (outlier, worst[False], worst[True]))
    # float32 itself carries 6e-8 per operand and ~1e-7 x sqrt(terms) per accumulation; 2e-6 is the bound the exact-f32
    # pipe is held to (test_gemm_vs_numpy).  With a 10^6 outlier the ordinary entries sit 2^-20 below the row scale and
    # keep ~19 significant bits: 4e-6.
    assert worst[False] < (2e-6 if outlier < 1e6 else 4e-6), worst
    assert worst[True] < (2e-6 if outlier < 1e6 else 4e-6), worst


def test_count_structure_is_detected_only_where_it_exists(engine):
    """X = counts / std (cnmf.py:546) has the structure (gemm_mode 4: f16 planes), also with a few counts above
    2048 (second plane); the same matrix with one entry nudged off the integer grid, or with a count above 65 535,
    does not (the general path: X itself as two f16 planes with a per-row exponent, 5)."""
    C, _ = synth.topic_counts(1024, 520, 6, 5.0, 0.3, 2)
    C = C[:, C.sum(axis=0) > 0]
    C = C[C.sum(axis=1) > 0]
    X = (C / C.std(axis=0, ddof=1)).astype(np.float64)
    ks, seeds = [9] * 29, list(range(1, 30))
    results = {}
    for tag in ("counts", "big", "nudged", "huge"):
        Xt = X.copy()
        if tag == "nudged":
            i, g = np.argwhere(C > 0)[0]
            Xt[i, g] *= 1.37
        if tag == "big":                                  # counts of 300, 2500 and 5000 in three places
            for (i, g, c) in ((3, 5, 300), (700, 400, 2500), (11, 17, 5000)):
                Xt[i, g] = X[:, g][X[:, g] > 0].min() * c / C[:, g][C[:, g] > 0].min()
        if tag == "huge":
            Xt[3, 5] = X[:, 5][X[:, 5] > 0].min() * 70000
        engine.set_matrix(Xt)
        H, _, n_iter, _ = engine.nmf_batch(ks, seeds=seeds, max_iter=20, warn=False)
        assert engine.last_stats["kc"] == 256
        results[tag] = (engine.last_stats["gemm_mode"], H)
    assert [results[t][0] for t in ("counts", "big", "nudged", "huge")] == [4, 4, 5, 5]      # 5: any X as two f16 planes
    # and the count path (with and without the second plane) computes the same factorisation as the exact-f32 pipe
    import os
    for tag in ("counts", "big"):
        Xt = X.copy()
        if tag == "big":
            for (i, g, c) in ((3, 5, 300), (700, 400, 2500), (11, 17, 5000)):
                Xt[i, g] = X[:, g][X[:, g] > 0].min() * c / C[:, g][C[:, g] > 0].min()
        os.environ["CNMF_GEMM3"] = "0"
        try:
            engine.set_matrix(Xt)
            H0, _, _, _ = engine.nmf_batch(ks, seeds=seeds, max_iter=20, warn=False)
            assert engine.last_stats["gemm_mode"] == 0
        finally:
            del os.environ["CNMF_GEMM3"]
        for a, b in zip(results[tag][1], H0):
            assert np.abs(a - b).max() <= 1e-4 * max(1.0, np.abs(b).max()), tag
    # the detection can be switched off (Engine.set_count_detection / cnmf_set_count_detection): general path
    engine.set_count_detection(False)
    try:
        engine.set_matrix(X)
        Hn, _, _, _ = engine.nmf_batch(ks, seeds=seeds, max_iter=20, warn=False)
        assert engine.last_stats["gemm_mode"] == 5
        for a, b in zip(results["counts"][1], Hn):
            assert np.abs(a - b).max() <= 1e-4 * max(1.0, np.abs(b).max())
    finally:
        engine.set_count_detection(True)


def test_general_matrix_on_the_f16_pipe_matches_oracle(engine):
    """gemm_mode 5 (the default for any matrix that is NOT count-structured: Harmony-corrected, TPM-normalised ...): X
    and X^T as two f16 planes of x * 2^s_row each, the factor likewise, all four plane products (4 MFMAs), the per-row
    exponent undone per output column in the GEMM epilogue.  Every restart of a full-width batch against its float64
    oracle run, on a matrix with a wide dynamic range per row and per column; the product itself against float64."""
    rs = np.random.RandomState(17)
    X64 = synth.make_config("C1", dtype=np.float64)
    X64 = X64 * np.exp(0.8 * rs.standard_normal((X64.shape[0], 1))) * np.exp(0.8 * rs.standard_normal((1, X64.shape[1])))
    X64 += 0.01 * np.abs(rs.standard_normal(X64.shape))          # nothing integer-like about it
    engine.set_matrix(X64)
    ks = [int(k) for k in rs.randint(5, 10, size=44)]
    seeds = [int(s) for s in rs.randint(1, 2**31 - 1, size=44)]
    H, _, n_iter, viol = engine.nmf_batch(ks, seeds=seeds)
    assert engine.last_stats["kc"] == 256 and engine.last_stats["gemm_mode"] == 5
    for k, seed, h, n in list(zip(ks, seeds, H, n_iter))[::3]:
        _, H_ref, n_ref = nmf_cd.nmf(X64, k, seed=seed)
        _check(H_ref, n_ref, h, n, slack=3)
    H2, _, n2, _ = engine.nmf_batch(ks, seeds=seeds)
    assert list(n2) == list(n_iter) and all(np.array_equal(a, b) for a, b in zip(H, H2))
    # against the 3 x 3 bf16 planes of rounds 1-2 (CNMF_G2G=0 is read once per process: compared through the oracle only)


def test_general_path_spread_stream_is_bit_identical_to_the_burst_loop(engine, monkeypatch):
    """Round 6: the general-matrix GEMMs (gemm_mode 5) issue their LDS-DMA pieces between the MFMA groups and read their
    fragments in two batches (the count kernels' spread stream, kernels_gemm2h.hip.h SPREADG) -- the MFMA order per accumulator
    is the burst loop's, so every bit of every result must be (CNMF_G2_GVAR=0: the burst loop).  A 256-column batch (pass B
    split-K, pass A stream-K with cut tiles) and a 1024-column one (four component groups, the identity XCD mapping), with
    refill, narrowing and partial tiles in the tail."""
    X = synth.make_config("C3", dtype=np.float32, n_cells=9000)
    rs = np.random.RandomState(5)
    X = (X * np.exp(0.3 * rs.standard_normal((X.shape[0], 1)))).astype(np.float32) + np.float32(0.003)
    engine.set_matrix(X)
    for n_restarts, kc in ((40, 256), (150, 1024)):
        ks = [int(k) for k in rs.randint(5, 14, size=n_restarts)]
        seeds = [int(s) for s in rs.randint(1, 2**31 - 1, size=n_restarts)]
        H, _, n_iter, viol = engine.nmf_batch(ks, seeds=seeds, max_iter=60, warn=False, kc_max=kc)
        assert engine.last_stats["kc"] == kc and engine.last_stats["gemm_mode"] == 5, engine.last_stats
        monkeypatch.setenv("CNMF_G2_GVAR", "0")
        H0, _, n0, viol0 = engine.nmf_batch(ks, seeds=seeds, max_iter=60, warn=False, kc_max=kc)
        monkeypatch.delenv("CNMF_G2_GVAR")
        assert list(n0) == list(n_iter) and np.array_equal(viol, viol0)
        assert all(np.array_equal(a, b) for a, b in zip(H, H0))
        # the opt-in partial-tile passes of the tail (CNMF_PART=1: dead 32-column tiles are neither multiplied nor stored) on
        # the same stream: what a live restart computes does not change
        monkeypatch.setenv("CNMF_PART", "1")
        Hp, _, n_p, violp = engine.nmf_batch(ks, seeds=seeds, max_iter=60, warn=False, kc_max=kc)
        monkeypatch.delenv("CNMF_PART")
        assert list(n_p) == list(n_iter) and all(np.array_equal(a, b) for a, b in zip(H, Hp))
    _, H_ref, _ = nmf_cd.nmf(X.astype(np.float64), ks[0], seed=seeds[0], max_iter=60)
    maxabs, relfro = nmf_cd.spectra_error(H_ref, H[0])
    assert maxabs <= 1e-4 and relfro <= 1e-3, (maxabs, relfro)


@pytest.mark.parametrize("g3mode", ["0", "1", "2", "3", "4"])
def test_full_width_batch_matches_oracle_in_every_gemm_mode(engine, monkeypatch, g3mode):
    """256 packed columns (the width at which the split-operand GEMM takes over): every restart
    against its independent float64 oracle run, for the exact-f32 pipe (0), both general split-operand
    variants (1, 2), the count-structured path on three bf16 planes (3) and on two f16 planes (4, the
    default).  Same tolerance in all five."""
    monkeypatch.setenv("CNMF_GEMM3", g3mode)
    X64 = synth.make_config("C1", dtype=np.float64)
    engine.set_matrix(X64)
    rs = np.random.RandomState(11)
    ks = [int(k) for k in rs.randint(5, 10, size=44)]
    seeds = [int(s) for s in rs.randint(1, 2**31 - 1, size=44)]
    assert sum(ks) > 256
    H, _, n_iter, viol = engine.nmf_batch(ks, seeds=seeds)
    assert engine.last_stats["kc"] == 256 and engine.last_stats["gemm_mode"] == min(int(g3mode), 4)
    for k, seed, h, n in list(zip(ks, seeds, H, n_iter))[::3]:
        _, H_ref, n_ref = nmf_cd.nmf(X64, k, seed=seed)
        _check(H_ref, n_ref, h, n, slack=3)
    assert (viol[n_iter < 1000] <= 1e-4).all()
    H2, _, n2, _ = engine.nmf_batch(ks, seeds=seeds)
    assert list(n2) == list(n_iter) and all(np.array_equal(a, b) for a, b in zip(H, H2))


@pytest.mark.parametrize("g3mode,kc", [("4", 512), ("4", 768), ("3", 512), ("2", 512), ("4", 1024), ("5", 512), ("5", 1024)])
def test_wide_batch_512_columns_matches_oracle(engine, monkeypatch, g3mode, kc):
    """Wide batches (512 / 768 packed columns = 2 / 3 component groups per GEMM pass: the default for jobs of >= 1536
    columns on large matrices): pass A walks its tiles component-group-major, pass B spreads (row tile, group, K split)
    over the XCDs, the cut flags are looked up by (row tile, group) -- every restart against its float64 oracle run, on
    the count path (f16 and bf16 planes) and the general split-operand path; then the tail narrows the batch in steps
    of 256 columns on the same kernels (more restarts than fit at once, all of different length)."""
    monkeypatch.setenv("CNMF_GEMM3", min(g3mode, "4"))
    X64 = synth.make_config("C1", dtype=np.float64)
    rs = np.random.RandomState(12)
    if g3mode == "5":                 # any matrix that is not count-structured: X itself as two f16 planes
        X64 = X64 * np.exp(0.5 * rs.standard_normal((X64.shape[0], 1))) + 0.01 * np.abs(rs.standard_normal(X64.shape))
    engine.set_matrix(X64)
    n = 150 if kc < 1024 else 200
    ks = [int(k) for k in rs.randint(5, 10, size=n)]
    seeds = [int(s) for s in rs.randint(1, 2**31 - 1, size=n)]
    assert sum(ks) > kc + 256
    H, _, n_iter, viol = engine.nmf_batch(ks, seeds=seeds, kc_max=kc)
    assert engine.last_stats["kc"] == kc and engine.last_stats["gemm_mode"] == int(g3mode)
    for k, seed, h, it in list(zip(ks, seeds, H, n_iter))[::7]:
        _, H_ref, n_ref = nmf_cd.nmf(X64, k, seed=seed)
        _check(H_ref, n_ref, h, it, slack=3)
    assert (viol[n_iter < 1000] <= 1e-4).all()
    H2, _, n2, _ = engine.nmf_batch(ks, seeds=seeds, kc_max=kc)
    assert list(n2) == list(n_iter) and all(np.array_equal(a, b) for a, b in zip(H, H2))     # deterministic
    # the same restarts in a 256-wide batch: same answers to rounding (other split-K / stream-K partitions)
    H3, _, n3, _ = engine.nmf_batch(ks[:40], seeds=seeds[:40], kc_max=256)
    for a, b, na, nb in zip(H[:40], H3, n_iter[:40], n3):
        assert abs(int(na) - int(nb)) <= max(2, int(nb) // 100)
        maxabs, relfro = nmf_cd.spectra_error(b, a)
        assert maxabs <= 1e-4 and relfro <= 1e-3


def test_device_standard_normal_matches_numpy(engine):
    """numpy RandomState(seed).standard_normal reproduced on the device (MT19937 +
    legacy polar gauss); known-answer vector from SURVEY.md 8c first."""
    z = engine.debug_standard_normal(59886188, 3)
    assert np.allclose(z, [0.29526446, 0.80487632, -0.3867717], atol=1e-8)
    for seed, n in [(1, 10), (59886188, 5001), (2**31 - 2, 100000), (1812018521, 1249)]:
        ref = np.random.RandomState(seed).standard_normal(n)
        z = engine.debug_standard_normal(seed, n)
        # bit-exact except (rarely) 1 ulp from log(); never more than a few ulp
        assert np.max(np.abs(z - ref) / np.maximum(np.abs(ref), 1e-300)) < 1e-14
        assert np.mean(z == ref) > 0.5


def test_single_restart_custom_init_C1(engine):
    X64 = synth.make_config("C1", dtype=np.float64)
    engine.set_matrix(X64)
    k, seed = 7, 59886188
    W0, H0 = nmf_cd.random_init(X64, k, seed)
    W_ref, H_ref, n_ref = nmf_cd.nmf(X64, k, W0=W0, H0=H0)
    H, W, n_iter, viol = engine.nmf_batch([k], W0=[W0], H0=[H0], return_W=True)
    _check(H_ref, n_ref, H[0], n_iter[0])
    # usages: same component order (same init), compare directly after scaling
    assert np.abs(W[0] - W_ref).max() <= 1e-3 * np.abs(W_ref).max()
    assert viol[0] <= 1e-4


def test_seeded_restarts_match_sklearn_C1(engine):
    """init='random' generated on the device from the ledger seeds vs. scikit-learn itself."""
    from sklearn.decomposition import non_negative_factorization
    X64 = synth.make_config("C1", dtype=np.float64)
    engine.set_matrix(X64)
    ks = [7, 7, 5, 9, 6]
    seeds = [59886188, 1812018521, 1173234957, 12345, 7]
    H, _, n_iter, _ = engine.nmf_batch(ks, seeds=seeds)
    for k, seed, h, n in zip(ks, seeds, H, n_iter):
        W_ref, H_ref, n_ref = non_negative_factorization(
            X64, n_components=k, init="random", solver="cd", beta_loss="frobenius", tol=1e-4,
            max_iter=1000, random_state=seed, alpha_W=0.0, alpha_H=0.0, l1_ratio=0.0)
        _check(H_ref, n_ref, h, n, slack=3)


def test_batch_refill_many_restarts(engine):
    """More restarts than fit in the packed columns: slots are retired and refilled;
    every restart must still equal its independent oracle run."""
    X64 = synth.make_config("C1", dtype=np.float64, n_cells=600)
    engine.set_matrix(X64)
    rs = np.random.RandomState(3)
    ks = [int(k) for k in rs.randint(3, 12, size=24)]
    seeds = [int(s) for s in rs.randint(1, 2**31 - 1, size=24)]
    H, _, n_iter, _ = engine.nmf_batch(ks, seeds=seeds, kc_max=64)
    assert engine.last_stats["kc"] == 64
    for k, seed, h, n in zip(ks, seeds, H, n_iter):
        _, H_ref, n_ref = nmf_cd.nmf(X64, k, seed=seed)
        _check(H_ref, n_ref, h, n, slack=3)


def test_regularised_restart(engine):
    X64 = synth.make_config("C1", dtype=np.float64, n_cells=500)
    engine.set_matrix(X64)
    k, seed = 6, 42
    _, H_ref, n_ref = nmf_cd.nmf(X64, k, seed=seed, alpha_W=0.002, alpha_H=0.001, l1_ratio=0.3)
    H, _, n_iter, _ = engine.nmf_batch([k], seeds=[seed], alpha_W=0.002, alpha_H=0.001, l1_ratio=0.3)
    _check(H_ref, n_ref, H[0], n_iter[0], slack=3)


def test_nnls_refit(engine):
    X64 = synth.make_config("C1", dtype=np.float64)
    engine.set_matrix(X64)
    _, H, _ = nmf_cd.nmf(X64, 7, seed=11)
    Hn = H / H.sum(axis=1, keepdims=True)
    W_ref, n_ref = nmf_cd.nnls(X64, Hn)
    W, n = engine.nnls(Hn)
    assert abs(n - n_ref) <= 2
    assert np.abs(W - W_ref).max() <= 1e-3 * np.abs(W_ref).max()


def test_csr_upload_equals_dense(engine):
    import scipy.sparse as sp
    X = synth.make_config("C1", dtype=np.float32, n_cells=300)
    X[X < 1.0] = 0
    X = X[X.sum(axis=1) > 0]
    engine.set_matrix(X)
    Hd, _, nd, _ = engine.nmf_batch([5], seeds=[9])
    engine.set_matrix(sp.csr_matrix(X))
    Hs, _, ns, _ = engine.nmf_batch([5], seeds=[9])
    # (X.mean() of a scipy matrix and of an ndarray differ in the last float32 bit, exactly as
    #  in sklearn, so the random init is scaled by an avg that differs by 1 ulp)
    assert abs(int(nd[0]) - int(ns[0])) <= 1
    assert np.allclose(Hd[0], Hs[0], rtol=1e-3, atol=1e-5)


def test_max_iter_warns(engine):
    from cnmf_amd.engine import ConvergenceWarning
    X64 = synth.make_config("C1", dtype=np.float64, n_cells=300)
    engine.set_matrix(X64)
    with pytest.warns(ConvergenceWarning):
        _, _, n_iter, _ = engine.nmf_batch([5], seeds=[1], max_iter=3)
    assert n_iter[0] == 3


def test_rank_above_kmax_is_rejected(engine):
    X64 = synth.make_config("C1", dtype=np.float64, n_cells=300)
    engine.set_matrix(X64)
    with pytest.raises(NotImplementedError):
        engine.nmf_batch([129], seeds=[1])


def test_negative_input_raises(engine):
    with pytest.raises(ValueError):
        engine.set_matrix(np.array([[1.0, -1.0], [0.5, 2.0]]))


def test_x_matmul_vs_numpy(engine):
    X = synth.make_config("C1", dtype=np.float64, n_cells=777)
    engine.set_matrix(X)
    rs = np.random.RandomState(0)
    for c in (1, 17, 33, 100):
        Q = rs.standard_normal((X.shape[1], c))
        ref = X @ Q
        assert np.abs(engine.x_matmul(Q) - ref).max() <= 2e-6 * np.abs(ref).max() * np.sqrt(X.shape[1])
        Q2 = rs.standard_normal((X.shape[0], c))
        ref2 = X.T @ Q2
        assert np.abs(engine.x_matmul(Q2, trans=True) - ref2).max() <= 2e-6 * np.abs(ref2).max() * np.sqrt(X.shape[0])


def test_nndsvd_long_matrix_takes_the_chunked_gram(engine):
    """9 000 cells: the Cholesky-QR normaliser of the range finder forms its Gram matrices from 64 chunks of positions
    (vectors of 8 192 positions and more) -- every restart of a group against scikit-learn's _initialize_nmf."""
    from sklearn.decomposition._nmf import _initialize_nmf
    X = synth.make_config("C1", dtype=np.float64, n_cells=9000)
    engine.set_matrix(X)
    ks, seeds = [5, 9, 13, 30], [3, 14, 15, 92]
    for (k, seed), (Wb, Hb) in zip(zip(ks, seeds), engine.nndsvd_init_batch(ks, seeds)):
        W_ref, H_ref = _initialize_nmf(X, k, init="nndsvd", random_state=seed)
        assert np.abs(Wb - W_ref).max() <= 1e-3 * np.abs(W_ref).max(), (k, seed)
        assert np.abs(Hb - H_ref).max() <= 1e-3 * np.abs(H_ref).max(), (k, seed)


@pytest.mark.parametrize("shape", ["tall", "wide"])
def test_nndsvd_init_matches_sklearn(engine, shape):
    """`--init nndsvd` (cnmf.py:1252): randomized-SVD products on the device, factorizations on the host."""
    from sklearn.decomposition._nmf import _initialize_nmf
    X = synth.make_config("C1", dtype=np.float64, n_cells=900 if shape == "tall" else 300)
    engine.set_matrix(X)
    W_ref, H_ref = _initialize_nmf(X, 6, init="nndsvd", random_state=42)
    W0, H0 = engine.nndsvd_init(6, random_state=42)
    assert np.abs(W0 - W_ref).max() <= 1e-3 * np.abs(W_ref).max()
    assert np.abs(H0 - H_ref).max() <= 1e-3 * np.abs(H_ref).max()
    # and a restart from it lands where sklearn's lands
    from sklearn.decomposition import non_negative_factorization
    Wr, Hr, nr = non_negative_factorization(X, n_components=6, init="nndsvd", solver="cd", tol=1e-4, max_iter=1000,
                                            random_state=42)
    H, _, n_iter, _ = engine.nmf_batch([6], W0=[W0], H0=[H0])
    _check(Hr, nr, H[0], n_iter[0], slack=3)
    # the batched form (range finders of several restarts side by side in ONE pass over X): every restart as sklearn's,
    # with the power iterations entirely on the device (Cholesky-QR normaliser) and with the host LU of round 2
    ks, seeds = [4, 6, 6, 9, 5, 30], [1, 42, 7, 3, 11, 5]
    for (k, seed), (Wb, Hb), (Wh, Hh) in zip(zip(ks, seeds), engine.nndsvd_init_batch(ks, seeds),
                                             engine.nndsvd_init_batch(ks, seeds, device_range_finder=False)):
        assert np.abs(Wh - Wb).max() <= 1e-3 * np.abs(Wb).max() and np.abs(Hh - Hb).max() <= 1e-3 * np.abs(Hb).max()
        W_ref, H_ref = _initialize_nmf(X, k, init="nndsvd", random_state=seed)
        assert Wb.shape == W_ref.shape and Hb.shape == H_ref.shape
        assert np.abs(Wb - W_ref).max() <= 1e-3 * np.abs(W_ref).max(), (k, seed)
        assert np.abs(Hb - H_ref).max() <= 1e-3 * np.abs(H_ref).max(), (k, seed)


@pytest.mark.parametrize("case", ["few_cells", "low_rank", "duplicated_rows"])
def test_nndsvd_rank_deficient_blocks(engine, case):
    """rank(X) < k + 10: the Cholesky-QR normaliser of the device range finder meets pivots that are zero up to round-off
    (round-3 advisor finding: a floored pivot compounded over several deficient columns overflowed R^-1 to inf / NaN and
    factorize aborted where scikit-learn's pivoted LU succeeds).  The dependent directions are dropped; every component
    scikit-learn determines (singular value > 0, i.e. up to rank(X)) comes out as scikit-learn's."""
    from sklearn.decomposition._nmf import _initialize_nmf
    rs = np.random.RandomState(5)
    if case == "few_cells":                      # min(N, G) = 12 < k + 10 = 18
        X, k, rank = rs.gamma(0.6, 1.0, size=(12, 200)), 8, 12
    elif case == "low_rank":                     # noiseless rank-5 matrix, k + 10 = 14 columns
        X, k, rank = rs.gamma(1.0, 1.0, size=(300, 5)) @ rs.gamma(1.0, 1.0, size=(5, 200)), 4, 5
    else:                                        # 40 distinct cells, each 10 times
        X, k, rank = np.repeat(rs.gamma(0.6, 1.0, size=(40, 150)), 10, axis=0), 35, 40
    engine.set_matrix(X)
    W0, H0 = engine.nndsvd_init(k, random_state=7)
    assert np.isfinite(W0).all() and np.isfinite(H0).all()
    W_ref, H_ref = _initialize_nmf(X, k, init="nndsvd", random_state=7)
    assert W0.shape == W_ref.shape and H0.shape == H_ref.shape
    assert k <= rank
    assert np.abs(W0 - W_ref).max() <= 2e-3 * np.abs(W_ref).max()
    assert np.abs(H0 - H_ref).max() <= 2e-3 * np.abs(H_ref).max()
    # and the restart from it runs (the failure mode was a ValueError out of the host SVD)
    H, _, n_iter, _ = engine.nmf_batch([k], W0=[W0], H0=[H0])
    assert np.isfinite(H[0]).all() and n_iter[0] >= 1


@pytest.mark.parametrize("case", ["size_factors_1pct", "size_factors_3e-3", "log1p", "tpm_like", "gene_jitter"])
def test_near_grid_matrices_are_not_snapped(engine, case):
    """Matrices that are count-DERIVED but no longer (integer x one constant per gene) must take the general path
    (gemm_mode 5), not be snapped onto the count grid (round-3 review, weak #10): counts scaled by per-CELL size factors
    close to 1 (1 % and 0.3 % spread: deviations of several 1e-3 count units on the larger counts), log1p-transformed
    counts, counts-per-10k (TPM-like), and per-entry multiplicative jitter of 0.5 %.  And the result on that path is the
    float64 oracle's for the matrix AS GIVEN."""
    C, _ = synth.topic_counts(1024, 520, 6, 5.0, 0.3, 2)
    C = C[:, C.sum(axis=0) > 0]
    C = C[C.sum(axis=1) > 0].astype(np.float64)
    rs = np.random.RandomState(11)
    if case.startswith("size_factors"):
        spread = 1e-2 if case.endswith("1pct") else 3e-3
        Xc = C * (1.0 + spread * (rs.rand(C.shape[0], 1) - 0.5) * 2)
    elif case == "log1p":
        Xc = np.log1p(C)
    elif case == "tpm_like":
        Xc = C / C.sum(axis=1, keepdims=True) * 1e4
    else:
        Xc = C * (1.0 + 5e-3 * (rs.rand(*C.shape) - 0.5) * 2)
    X = Xc / Xc.std(axis=0, ddof=1)
    engine.set_matrix(X)
    ks, seeds = [9] * 29, list(range(1, 30))
    H, _, n_iter, _ = engine.nmf_batch(ks, seeds=seeds, max_iter=25, warn=False)
    assert engine.last_stats["kc"] == 256 and engine.last_stats["gemm_mode"] == 5, (case, engine.last_stats["gemm_mode"])
    for r in (0, 7, 28):
        _, H_ref, n_ref = nmf_cd.nmf(X, ks[r], seed=seeds[r], max_iter=25)
        maxabs, relfro = nmf_cd.spectra_error(H_ref, H[r])
        assert maxabs <= 1e-4 and relfro <= 1e-3, (case, r, maxabs, relfro)
