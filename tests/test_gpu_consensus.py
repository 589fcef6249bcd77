"""GPU parity tests of the consensus core (cnmf.py:871-936) through the C-ABI, against
the numpy oracle (oracle/consensus.py, pinned to sklearn/pandas) and against the fixtures
written by the unmodified reference (tests/golden/ref_small.npz).

Bars: index outputs (density filter, k-means labels) bit-exact; float64 outputs within
1e-9 (they are float64 on the device too); the reference's own bar for consensus spectra is
sum of squared differences < 1e-4 (tests/test_reproducibility.py:12)."""
import os

import numpy as np
import pytest

from cnmf_amd import synth
from oracle import consensus as oc
from oracle import nmf_cd

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "ref_small.npz")


def _same_partition(a, b):
    m = {}
    for x, y in zip(a, b):
        if m.setdefault(int(x), int(y)) != int(y):
            return False
    return len(set(m.values())) == len(m)


def test_consensus_core_vs_oracle(engine):
    S, _ = synth.consensus_stress(R=640, G=300, k=8, n_outliers=24, seed=1)
    rs = np.random.RandomState(0)
    X = np.abs(rs.standard_normal((50, 300)))
    ref = oc.consensus_core(S, X, 8, density_threshold=0.5)
    out = engine.consensus(S, 8, density_threshold=0.5, return_dist=True)
    assert np.abs(out["topics_dist"] - ref["topics_dist"]).max() < 1e-7     # sqrt amplifies 1e-16 near 0
    assert np.abs(out["local_density"] - ref["local_density"]).max() < 1e-9
    assert np.array_equal(out["density_filter"], ref["density_filter"])
    kept = out["density_filter"]
    assert (out["labels"][~kept] == -1).all()
    assert np.array_equal(out["labels"][kept] + 1, ref["kmeans_labels"])     # same seeds -> same label ids
    assert np.abs(out["median_spectra"] - ref["median_spectra"]).max() < 1e-12
    assert abs(out["inertia"] - ref["inertia"]) <= 1e-9 * ref["inertia"]


def test_consensus_unstructured_spectra_same_kmeans(engine):
    """No cluster structure: every k-means++ draw and Lloyd tie matters."""
    rs = np.random.RandomState(4)
    S = np.abs(rs.standard_normal((300, 64)))
    l2 = oc.l2_normalise(S)
    labels, _, inertia = oc.kmeans(l2, 7)
    out = engine.consensus(S, 7, skip_density=True, want_silhouette=True)
    assert np.array_equal(out["labels"], labels)
    assert abs(out["inertia"] - inertia) <= 1e-9 * inertia
    assert abs(out["silhouette"] - oc.silhouette_score(l2, labels)) < 1e-9


def test_consensus_zero_rows_after_filter_raises(engine):
    S, _ = synth.consensus_stress(R=200, G=100, k=4, n_outliers=10, seed=2)
    with pytest.raises(RuntimeError, match="Zero components remain"):
        engine.consensus(S, 4, density_threshold=1e-9)


@pytest.mark.parametrize("k,thr", [(5, 0.5), (4, 2.0)])
def test_consensus_golden_reference(engine, k, thr):
    """Merged spectra written by the unmodified reference -> its consensus files."""
    g = dict(np.load(GOLD, allow_pickle=False))
    X = g["norm_counts"]
    out = engine.consensus(g["merged_k%d" % k], k, density_threshold=thr)
    assert np.abs(out["local_density"] - g["local_density_k%d" % k]).max() < 1e-9
    engine.set_matrix(X)
    W, _ = engine.nnls(out["median_spectra"])
    norm = W / W.sum(axis=1, keepdims=True)
    order = np.argsort(-norm.sum(axis=0), kind="stable")                    # cnmf.py:939-946
    med = out["median_spectra"][order]
    assert ((med - g["consensus_spectra_k%d" % k]) ** 2).sum() < 1e-4      # the reference's TOLERANCE


@pytest.mark.parametrize("k", [4, 5, 6])
def test_stats_mode_golden_reference(engine, k):
    """k_selection_plot's per-k statistics (cnmf.py:922-936): silhouette + prediction error."""
    g = dict(np.load(GOLD, allow_pickle=False))
    X = g["norm_counts"]
    engine.set_matrix(X)
    out = engine.consensus(g["merged_k%d" % k], k, skip_density=True, want_silhouette=True)
    W, _ = engine.nnls(out["median_spectra"])
    err = engine.prediction_error(W, out["median_spectra"])
    _, _, sil_ref, err_ref = g["stats_k%d" % k]
    assert abs(out["silhouette"] - sil_ref) < 1e-8
    assert abs(err - err_ref) <= 2e-5 * err_ref          # X and W are float32 on the device


def test_prediction_error_vs_numpy(engine):
    X = synth.make_config("C1", dtype=np.float64, n_cells=333)
    engine.set_matrix(X)
    W, H, _ = nmf_cd.nmf(X, 6, seed=3)
    ref = ((X - W @ H) ** 2).sum()
    assert abs(engine.prediction_error(W, H) - ref) <= 1e-6 * ref


def test_consensus_stress_C5_full_size_vs_oracle(engine):
    """BASELINE config 5 at FULL size (5000 x 2000, k=20) against oracle/consensus.py (the numpy restatement pinned to
    sklearn / pandas in tests/test_oracle_consensus.py): density to 1e-9, filter and k-means labels exact, medians to
    1e-12, inertia to 1e-9 -- plus the size-independent properties."""
    S, truth = synth.consensus_stress(R=5000, G=2000, k=20, n_outliers=100, seed=0)
    out = engine.consensus(S, 20, density_threshold=0.5)
    ref = oc.consensus_core(S, np.abs(np.random.RandomState(0).standard_normal((20, 2000))), 20, density_threshold=0.5)
    assert np.abs(out["local_density"] - ref["local_density"]).max() < 1e-9
    assert np.array_equal(out["density_filter"], ref["density_filter"])
    kept = out["density_filter"]
    assert (out["labels"][~kept] == -1).all()
    assert np.array_equal(out["labels"][kept] + 1, ref["kmeans_labels"])
    assert np.abs(out["median_spectra"] - ref["median_spectra"]).max() < 1e-12
    assert abs(out["inertia"] - ref["inertia"]) <= 1e-9 * ref["inertia"]
    assert kept.sum() == 4900 and not kept[truth < 0].any()              # exactly the noise rows go
    assert _same_partition(out["labels"][kept], truth[kept])             # the planted clusters come back
    assert np.allclose(out["median_spectra"].sum(axis=1), 1.0, atol=1e-12)
    assert (out["median_spectra"] >= 0).all()


@pytest.mark.parametrize("R,k", [(2600, 1), (2600, 2), (900, 3)])
def test_median_paths_large_clusters(engine, R, k):
    """Cluster sizes above 512 / 2048 members take the wider / the re-reading median kernels."""
    S, _ = synth.consensus_stress(R=R, G=96, k=k, n_outliers=0, seed=5)
    ref = oc.consensus_core(S, np.abs(np.random.RandomState(0).standard_normal((20, 96))), k, density_threshold=2.0)
    out = engine.consensus(S, k, density_threshold=2.0)
    assert np.array_equal(out["labels"] + 1, ref["kmeans_labels"])
    assert np.abs(out["median_spectra"] - ref["median_spectra"]).max() < 1e-12


def test_large_R_fallback_kernels_agree_with_the_register_kernels(engine, monkeypatch):
    """More than 8 192 kept spectra seed k-means++ from global memory (pp_fused_kernel) and more than 20 480 select the
    neighbours by re-reading the row (knn_density_kernel); both are forced here on a case the register kernels handle:
    the density must be bit-identical, labels / medians the same, and both must match the oracle."""
    S, _ = synth.consensus_stress(R=1500, G=120, k=7, n_outliers=40, seed=11)
    ref = oc.consensus_core(S, np.abs(np.random.RandomState(0).standard_normal((20, 120))), 7, density_threshold=0.5)
    fast = engine.consensus(S, 7, density_threshold=0.5)
    monkeypatch.setenv("CNMF_PP_GLOBAL", "1")
    monkeypatch.setenv("CNMF_KNN_GLOBAL", "1")
    slow = engine.consensus(S, 7, density_threshold=0.5)
    assert np.array_equal(fast["local_density"], slow["local_density"])
    for out in (fast, slow):
        assert np.abs(out["local_density"] - ref["local_density"]).max() < 1e-9
        assert np.array_equal(out["density_filter"], ref["density_filter"])
        assert np.array_equal(out["labels"][out["density_filter"]] + 1, ref["kmeans_labels"])
        assert np.abs(out["median_spectra"] - ref["median_spectra"]).max() < 1e-12


def test_kmeans_needs_several_lloyd_batches(engine):
    """Overlapping clusters: Lloyd runs for many iterations (the device applies the stopping rule; the host looks once per
    batch of three) and some inits stop on the tolerance -- iteration count, labels and inertia against the oracle."""
    rng = np.random.RandomState(3)
    centers = np.abs(rng.standard_normal((6, 40)))
    S = np.abs(centers[rng.randint(0, 6, 1200)] + 0.9 * rng.standard_normal((1200, 40))) + 1e-3
    ref = oc.consensus_core(S, np.abs(rng.standard_normal((20, 40))), 6, density_threshold=2.0)
    out = engine.consensus(S, 6, density_threshold=2.0)
    assert out["kmeans_n_iter"] == 13                      # KMeans(n_clusters=6, n_init=10, random_state=1).n_iter_ on this input
    assert np.array_equal(out["labels"] + 1, ref["kmeans_labels"])
    assert abs(out["inertia"] - ref["inertia"]) <= 1e-9 * ref["inertia"]


def test_more_than_8192_kept_spectra_take_the_global_memory_seeding(engine):
    """9 000 merged spectra: the KNN selection holds 36 values per thread (the 40-value instantiation), 8 850 rows
    survive the filter -- above the 8 192 the register k-means++ kernel holds -- so the seeding runs from global memory
    without being forced to; clusters of 1 475 members take the wide median kernel.  All against the oracle."""
    S, truth = synth.consensus_stress(R=9000, G=48, k=6, n_outliers=150, seed=3)
    ref = oc.consensus_core(S, np.abs(np.random.RandomState(0).standard_normal((20, 48))), 6, density_threshold=0.5)
    out = engine.consensus(S, 6, density_threshold=0.5)
    kept = out["density_filter"]
    assert kept.sum() == 8850 and np.array_equal(kept, ref["density_filter"])
    assert np.abs(out["local_density"] - ref["local_density"]).max() < 1e-9
    assert np.array_equal(out["labels"][kept] + 1, ref["kmeans_labels"])
    assert np.abs(out["median_spectra"] - ref["median_spectra"]).max() < 1e-12
    assert abs(out["inertia"] - ref["inertia"]) <= 1e-9 * ref["inertia"]


@pytest.mark.parametrize("R", [1024, 1025, 4097, 8192, 8193])
def test_kept_row_counts_at_the_edges_of_the_register_kernels(engine, R):
    """k-means++ holds 1 / 2 / 4 / 8 rows per thread of a 1024-thread workgroup and falls back to global memory above
    8 192; the KNN selection holds 4 ... 40 values per thread of 256: row counts on both sides of those edges (all rows
    kept: density_threshold 2.0), labels and medians against the oracle."""
    S, _ = synth.consensus_stress(R=R, G=24, k=4, n_outliers=0, seed=R)
    ref = oc.consensus_core(S, np.abs(np.random.RandomState(0).standard_normal((20, 24))), 4, density_threshold=2.0)
    out = engine.consensus(S, 4, density_threshold=2.0)
    assert out["density_filter"].all()
    assert np.abs(out["local_density"] - ref["local_density"]).max() < 1e-9
    assert np.array_equal(out["labels"] + 1, ref["kmeans_labels"])
    assert np.abs(out["median_spectra"] - ref["median_spectra"]).max() < 1e-12
    assert abs(out["inertia"] - ref["inertia"]) <= 1e-9 * ref["inertia"]
