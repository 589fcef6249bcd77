"""End-to-end drop-in test: the host-side mirror of the reference's cNMF object
(cnmf_amd/cnmf.py) run on the GPU against what the UNMODIFIED reference produced for the
same normalised matrix, ledger seed and parameters (tests/golden/ref_small.npz):

    prepare_from_matrix -> factorize (2 workers) -> combine -> k_selection_stats -> consensus

The reference's own test bar is sum of squared differences < 1e-4
(/root/reference/tests/test_reproducibility.py:12)."""
import errno
import os

import numpy as np
import pandas as pd
import pytest

from cnmf_amd.cnmf import cNMF, load_df_from_npz
from oracle import nmf_cd

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "ref_small.npz")
TOLERANCE = 1e-4


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(GOLD, allow_pickle=False))


@pytest.fixture(scope="module")
def run(tmp_path_factory, gold, engine):
    g = gold
    out = tmp_path_factory.mktemp("cnmf_run")
    obj = cNMF(output_dir=str(out), name="t", engine=engine)
    nc = pd.DataFrame(g["norm_counts"], index=["c%d" % i for i in range(g["norm_counts"].shape[0])],
                      columns=list(g["genes"]))
    tpm = pd.DataFrame(g["tpm"], index=nc.index, columns=list(g["tpm_genes"]))
    obj.prepare_from_matrix(nc, components=[4, 5, 6], n_iter=12, seed=14, beta_loss="frobenius", tpm=tpm)
    return obj


def test_ledger_and_yaml_match_reference(run, gold):
    led = load_df_from_npz(run.paths["nmf_replicate_parameters"])
    assert np.array_equal(led[["n_components", "iter", "nmf_seed"]].values.astype(np.int64), gold["ledger"])
    import yaml
    kw = yaml.load(open(run.paths["nmf_run_parameters"]), Loader=yaml.FullLoader)
    assert kw == dict(alpha_W=0.0, alpha_H=0.0, l1_ratio=0.0, beta_loss="frobenius", solver="cd",
                      tol=1e-4, max_iter=1000, init="random")


def test_factorize_shards_and_resume(run, gold):
    # worker 1 of 2 first: only its ledger rows get files (cnmf.py:52-53)
    run.factorize(worker_i=1, total_workers=2)
    led = load_df_from_npz(run.paths["nmf_replicate_parameters"])
    have = [os.path.exists(run.paths["iter_spectra"] % (int(r.n_components), int(r.iter))) for r in led.itertuples()]
    assert have == [(i - 1) % 2 == 0 for i in range(len(led))]
    with pytest.raises(FileNotFoundError) as ei:
        run.combine_nmf(4)
    assert ei.value.errno == errno.ENOENT
    # resume: mark completed, run the rest with skip_completed_runs on a single worker
    run.update_nmf_iter_params()
    run.factorize(worker_i=0, total_workers=1, skip_completed_runs=True)
    run.combine()
    for k in (4, 5, 6):
        merged = load_df_from_npz(run.paths["merged_spectra"] % k)
        ref = gold["merged_k%d" % k]
        assert merged.shape == ref.shape
        assert list(merged.index[:k + 1]) == ["iter0_topic%d" % (t + 1) for t in range(k)] + ["iter1_topic1"]
        # per restart: same components in the same order (same seed, same init)
        for it in range(12):
            maxabs, relfro = nmf_cd.spectra_error(ref[it * k:(it + 1) * k], merged.values[it * k:(it + 1) * k])
            assert maxabs <= 5e-4 and relfro <= 1e-3, (k, it, maxabs, relfro)


def test_k_selection_stats_match_reference(run, gold):
    stats = run.k_selection_stats()
    for row, k in zip(stats.itertuples(), (4, 5, 6)):
        kk, thr, sil, err = gold["stats_k%d" % k]
        assert row.k == k
        assert abs(row.silhouette - sil) < 5e-3
        assert abs(row.prediction_error - err) <= 1e-3 * err


@pytest.mark.parametrize("k,thr", [(5, 0.5), (4, 2.0)])
def test_consensus_matches_reference(run, gold, k, thr):
    med, usages = run.consensus(k, density_threshold=thr)
    rep = str(thr).replace(".", "_")
    assert ((med.values - gold["consensus_spectra_k%d" % k]) ** 2).sum() < TOLERANCE
    assert ((usages.values - gold["consensus_usages_k%d" % k]) ** 2).sum() < TOLERANCE * usages.size
    tpm_sp = load_df_from_npz(run.paths["gene_spectra_tpm"] % (k, rep)).values
    ref = gold["gene_spectra_tpm_k%d" % k]
    assert np.abs(tpm_sp - ref).max() <= 2e-3 * np.abs(ref).max()
    score = load_df_from_npz(run.paths["gene_spectra_score"] % (k, rep)).values
    assert np.abs(score - gold["gene_spectra_score_k%d" % k]).max() < 5e-3
    assert os.path.exists(run.paths["consensus_spectra__txt"] % (k, rep))
    assert os.path.exists(run.paths["local_density_cache"] % k)


def test_zero_count_cell_raises_like_prepare(tmp_path, engine):
    X = np.ones((10, 6))
    X[3] = 0
    obj = cNMF(output_dir=str(tmp_path), name="z", engine=engine)
    with pytest.raises(Exception, match="Error: .* cells have zero counts of overdispersed genes.*"):
        obj.prepare_from_matrix(X, components=[2], n_iter=2, seed=1)


def test_nmf_callsite_dtype_contract(run, gold):
    """cNMF._nmf returns arrays of X's dtype and rejects an H of another dtype like sklearn."""
    X = gold["norm_counts"]
    spectra, usages = run._nmf(X, dict(n_components=4, random_state=5, tol=1e-4, max_iter=50, solver="cd",
                                       beta_loss="frobenius", alpha_W=0.0, alpha_H=0.0, l1_ratio=0.0, init="random"))
    assert spectra.shape == (4, X.shape[1]) and usages.shape == (X.shape[0], 4) and spectra.dtype == X.dtype
    with pytest.raises(TypeError):
        run._nmf(X, dict(H=spectra.astype(np.float32), update_H=False, n_components=4, solver="cd",
                         beta_loss="frobenius", tol=1e-4, max_iter=50))
    with pytest.raises(NotImplementedError):
        run._nmf(X, dict(n_components=4, random_state=5, solver="mu", beta_loss=0.5))     # generic beta: not built


def test_device_normalisation_matches_get_norm_counts(engine, tmp_path):
    """`norm_counts.X /= norm_counts.X.std(axis=0, ddof=1)` (cnmf.py:546) on the device: float64
    statistics, one rounding to float32; the zero-cell error of cnmf.py:550-554; and a factorize that
    reuses the resident matrix gives what the host-normalised pipeline gives."""
    from cnmf_amd import synth
    from cnmf_amd.cnmf import cNMF
    C, _ = synth.topic_counts(700, 300, 5, 1.0, 0.6, 3)
    C = C[:, C.sum(axis=0) > 0]
    C = C[C.sum(axis=1) > 0].astype(np.float64)
    ref64 = C / C.std(axis=0, ddof=1)
    engine.set_matrix(C.astype(np.float32))
    std, rs = engine.scale_genes_unit_variance()
    assert np.allclose(std, C.std(axis=0, ddof=1), rtol=1e-13, atol=0)
    got = engine.get_matrix()
    ref32 = ref64.astype(np.float32)
    assert np.mean(got == ref32) > 0.99999                                   # the odd element may round the other way
    assert np.abs(got.astype(np.float64) - ref64).max() <= np.abs(ref64).max() * 2.0 ** -23
    assert np.allclose(rs, ref32.astype(np.float64).sum(axis=1), rtol=1e-12)
    assert engine.x_dtype == np.float64 and abs(engine.x_mean - ref64.mean()) <= 1e-8 * ref64.mean()
    # a gene without variance cannot be scaled
    Cz = C.copy(); Cz[:, 7] = 3.0
    engine.set_matrix(Cz.astype(np.float32))
    with pytest.raises(ValueError):
        engine.scale_genes_unit_variance()
    # the mirror class: zero cells raise the reference's message; otherwise factorize reuses the upload
    Cbad = C.copy(); Cbad[5, :] = 0
    obj = cNMF(output_dir=str(tmp_path), name="dn")
    with pytest.raises(Exception, match="cells have zero counts of overdispersed genes"):
        obj.prepare_from_counts(Cbad, components=[4], n_iter=2, seed=3)
    obj = cNMF(output_dir=str(tmp_path), name="dn2")
    nc = obj.prepare_from_counts(C, components=[4, 5], n_iter=3, seed=14)
    assert np.array_equal(nc.values, got.astype(np.float64))
    key_before = obj._engine_key
    obj.factorize(write_iter_files=False)
    assert obj._engine_key == key_before                                      # no second upload
    obj2 = cNMF(output_dir=str(tmp_path), name="dn3")
    obj2.prepare_from_matrix(ref64, components=[4, 5], n_iter=3, seed=14)
    obj2.factorize(write_iter_files=False)
    for key in obj2.spectra_cache:
        a, b = np.asarray(obj.spectra_cache[key]), np.asarray(obj2.spectra_cache[key])
        assert np.abs(a - b).max() <= 1e-4 * max(1.0, np.abs(b).max())


def test_merged_spectra_served_from_the_device_store(tmp_path, gold, engine):
    """Round 4: factorize keeps the spectra in the engine's resident store (``nmf_batch(resident="keep")``); k selection and
    consensus of the same process gather their merged spectra ON THE DEVICE (``cnmf_kselect_stats_store`` /
    ``cnmf_consensus_store``) instead of uploading them -- with bit-identical results to the upload path, which a fresh
    object reading the files takes; the store survives the TPM upload in between; a new prepare voids it."""
    g = gold
    nc = pd.DataFrame(g["norm_counts"], index=["c%d" % i for i in range(g["norm_counts"].shape[0])], columns=list(g["genes"]))
    tpm = pd.DataFrame(g["tpm"], index=nc.index, columns=list(g["tpm_genes"]))
    a = cNMF(output_dir=str(tmp_path), name="dev", engine=engine)
    a.prepare_from_matrix(nc, components=[4, 5], n_iter=6, seed=14, beta_loss="frobenius", tpm=tpm)
    engine.spectra_reset()
    a.factorize()
    assert engine.spectra_rows == 6 * (4 + 5) and engine.spectra_genes == nc.shape[1]
    # the store holds exactly what came back to the host
    store = engine.spectra_fetch()
    for (k, it), (off, gen) in a._store_rows.items():
        assert gen == engine.store_gen and np.array_equal(store[off:off + k], a.spectra_cache[(k, it)].astype(np.float32))
    a.combine()
    calls = []
    real_cons, real_ksel = engine.consensus, engine.kselect_stats
    engine.consensus = lambda *x, **kw: (calls.append(("consensus", kw.get("store_rows") is not None)), real_cons(*x, **kw))[1]
    engine.kselect_stats = lambda *x, **kw: (calls.append(("kselect", kw.get("store_rows_by_k") is not None)), real_ksel(*x, **kw))[1]
    try:
        stats_a = a.k_selection_stats()
        med_a, use_a = a.consensus(5, density_threshold=0.5)         # uploads the TPM matrix behind the core ...
        med_a4, _ = a.consensus(4, density_threshold=2.0)            # ... and the store still serves the next k
        assert calls == [("kselect", True), ("consensus", True), ("consensus", True)], calls
        # a fresh object on the same directory: merged spectra from the files, uploaded
        calls.clear()
        b = cNMF(output_dir=str(tmp_path), name="dev", engine=engine)
        stats_b = b.k_selection_stats()
        med_b, use_b = b.consensus(5, density_threshold=0.5)
        med_b4, _ = b.consensus(4, density_threshold=2.0)
        assert [c[1] for c in calls] == [False, False, False], calls
    finally:
        engine.consensus, engine.kselect_stats = real_cons, real_ksel
    assert np.array_equal(stats_a.values, stats_b.values)
    assert np.array_equal(med_a.values, med_b.values) and np.array_equal(use_a.values, use_b.values)
    assert np.array_equal(med_a4.values, med_b4.values)
    # a repeated factorize of the same ledger replaces its rows instead of appending behind them, and a
    # batch call brings back its OWN rows only; a resumed worker (other ledger rows) appends
    first = {key: v.copy() for key, v in a.spectra_cache.items()}
    a.factorize()
    assert engine.spectra_rows == 6 * (4 + 5)
    assert all(np.array_equal(first[key], a.spectra_cache[key]) for key in first)
    a.factorize(worker_i=0, total_workers=2)                         # a subset of the keys the object holds: appended
    assert engine.spectra_rows == 6 * (4 + 5) + sum(k for i, (k, it) in enumerate((k, it) for k in (4, 5) for it in range(6)) if i % 2 == 0)
    np.testing.assert_array_equal(engine.spectra_fetch(5, 3), engine.spectra_fetch()[5:8])
    # a new prepare voids the store rows
    a.prepare_from_matrix(nc, components=[4], n_iter=2, seed=3, beta_loss="frobenius", tpm=tpm)
    assert a._store_rows == {} and engine.spectra_rows == 0
    with pytest.raises(ValueError):
        engine.consensus(None, 4, store_rows=[0, 1, 2, 3, 4, 5, 6, 7])      # nothing there any more
