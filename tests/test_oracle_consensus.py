"""Pin the numpy restatement of the consensus core (oracle/consensus.py) to the live
scikit-learn / pandas functions the reference calls (cnmf.py:882-936).  CPU only."""
import numpy as np
import pandas as pd
import pytest

from cnmf_amd import synth
from oracle import consensus as oc


@pytest.fixture(scope="module")
def spectra():
    S, lab = synth.consensus_stress(R=420, G=250, k=6, n_outliers=20, seed=0)
    return S


def test_l2_and_distances_equal_sklearn(spectra):
    from sklearn.metrics.pairwise import euclidean_distances
    df = pd.DataFrame(spectra)
    l2_ref = (df.T / np.sqrt((df ** 2).sum(axis=1))).T.values          # cnmf.py:882
    l2 = oc.l2_normalise(spectra)
    assert np.array_equal(l2, l2_ref)
    assert np.abs(oc.euclidean_distances(l2) - euclidean_distances(l2)).max() < 1e-12


def test_local_density_equals_reference_expression(spectra):
    from sklearn.metrics.pairwise import euclidean_distances
    l2 = oc.l2_normalise(spectra)
    D = euclidean_distances(l2)
    n = int(0.30 * spectra.shape[0] / 6)
    order = np.argpartition(D, n + 1)[:, :n + 1]                        # cnmf.py:893
    ref = D[np.arange(D.shape[0])[:, None], order].sum(1) / n           # cnmf.py:895-896
    assert np.abs(oc.local_density(D, n) - ref).max() < 1e-12


@pytest.mark.parametrize("structured", [True, False])
def test_kmeans_equals_sklearn(spectra, structured):
    from sklearn.cluster import KMeans
    if structured:
        X, k = oc.l2_normalise(spectra), 6
    else:
        X, k = np.abs(np.random.RandomState(4).standard_normal((300, 40))), 7
    km = KMeans(n_clusters=k, n_init=10, random_state=1).fit(X)         # cnmf.py:908-909
    labels, centers, inertia = oc.kmeans(X, k)
    assert np.array_equal(labels, km.labels_)
    assert abs(inertia - km.inertia_) <= 1e-9 * km.inertia_
    assert np.abs(centers - km.cluster_centers_).max() < 1e-10


def test_kmeans_uniform_stream_is_data_independent():
    """The k-means++ draws: 1 + (k-1)*(2+int(log k)) uniforms per init from RandomState(1)."""
    u = oc.kmeans_uniforms(6)
    assert u.shape == (10, 1 + 5 * (2 + int(np.log(6))))
    rng = np.random.RandomState(1)
    assert u[0, 0] == rng.random_sample()


def test_median_equals_pandas(spectra):
    l2 = oc.l2_normalise(spectra)
    labels = (np.arange(l2.shape[0]) * 7919 % 5) + 1
    ref = pd.DataFrame(l2).groupby(pd.Series(labels)).median()          # cnmf.py:913
    labs, med = oc.groupby_median(l2, labels)
    assert list(labs) == list(ref.index) and np.array_equal(med, ref.values)


def test_silhouette_equals_sklearn(spectra):
    from sklearn.metrics import silhouette_score
    l2 = oc.l2_normalise(spectra)
    labels, _, _ = oc.kmeans(l2, 6)
    ref = silhouette_score(l2, labels + 1, metric="euclidean")          # cnmf.py:923
    assert abs(oc.silhouette_score(l2, labels + 1) - ref) < 1e-12


def test_consensus_core_runs_and_filters(spectra):
    rs = np.random.RandomState(0)
    X = np.abs(rs.standard_normal((80, spectra.shape[1])))
    out = oc.consensus_core(spectra, X, 6, density_threshold=0.5)
    assert out["density_filter"].sum() == 400            # the 20 noise rows are filtered
    assert out["median_spectra"].shape == (6, spectra.shape[1])
    assert np.allclose(out["median_spectra"].sum(axis=1), 1.0)
    assert out["rf_usages"].shape == (80, 6)
    st = oc.consensus_core(spectra, X, 6, stats_mode=True)
    assert -1 <= st["silhouette"] <= 1 and st["prediction_error"] > 0
    with pytest.raises(RuntimeError, match="Zero components remain"):
        oc.consensus_core(spectra, X, 6, density_threshold=1e-6)
