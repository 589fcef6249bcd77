"""Device-backed stand-ins for the three scikit-learn names the reference's ``consensus`` body calls
(cnmf.py:15-18: ``KMeans`` :908, ``euclidean_distances`` :891/988, ``silhouette_score`` :923).

``integration/hip_backend.py`` (INTEGRATION.md Option B: a subclass of the UNMODIFIED reference class) swaps them in
for the duration of ``consensus``.  Every one of them runs on the device or raises: there is no scikit-learn fallback.  They live in the package -- not next to the subclass -- because they need nothing
from the reference tree: the GPU tests drive them against ``libcnmf_hip.so`` and compare with the live scikit-learn
functions they replace (tests/test_gpu_option_b_replay.py)."""
import numpy as np


class DeviceKMeans:
    """Stand-in for ``sklearn.cluster.KMeans`` inside ``consensus`` (cnmf.py:908-911): ``fit`` + ``labels_``.
    ``KMeans(n_clusters=k, n_init=10, random_state=1)`` -> ``Engine.consensus(skip_density=True)`` on the rows it is
    given (bit-identical labels, DESIGN.md section 4 "Consensus in float64")."""

    def __init__(self, engine, n_clusters, n_init=10, random_state=1, **kw):
        self._engine, self.n_clusters, self.n_init, self.random_state = engine, n_clusters, n_init, random_state

    def fit(self, X, y=None):
        vals = X.values if hasattr(X, "values") else np.asarray(X)
        out = self._engine.consensus(vals, self.n_clusters, skip_density=True, want_silhouette=True,
                                     random_state=self.random_state, n_init=self.n_init)
        self.labels_ = out["labels"].astype(np.int32)
        self.inertia_ = out["inertia"]
        self._silhouette = (vals.shape, out["silhouette"])
        return self


def device_euclidean_distances(engine, X, Y=None, **kw):
    """``euclidean_distances(l2_spectra)`` (cnmf.py:891, 988): ``cnmf_pairwise_distances`` -- the distance kernel of the
    consensus core on the rows as they are, nothing else (round 3 ran a whole k = 1 consensus behind it)."""
    if Y is not None:
        raise NotImplementedError("the device stand-in computes all-pairs distances of ONE matrix (cnmf.py:891)")
    if kw.get("squared"):
        raise NotImplementedError("squared=True is not what cnmf.py:891 / :988 ask for")
    vals = X.values if hasattr(X, "values") else np.asarray(X)
    return engine.pairwise_distances(vals)[0]


def device_silhouette_score(engine, last_kmeans, X, labels, metric="euclidean", **kw):
    """``silhouette_score(l2_spectra, labels, metric='euclidean')`` (cnmf.py:923).  The fit of cnmf.py:909 already produced
    it on the device for exactly these rows and its own labels; any other rows / labels are scored by
    ``cnmf_pairwise_distances`` -- on the device too.  There is NO scikit-learn behind this (round-3 review, weak #9):
    a metric other than the reference's raises."""
    if metric != "euclidean" or kw.get("sample_size") is not None:
        raise NotImplementedError("the device stand-in scores metric='euclidean' on all samples (cnmf.py:923)")
    vals = X.values if hasattr(X, "values") else np.asarray(X)
    lab = np.asarray(labels.values if hasattr(labels, "values") else labels)
    km = last_kmeans
    # (the reference hands over labels + 1, cnmf.py:910-911: the same partition under another naming)
    if (km is not None and km._silhouette[0] == vals.shape and lab.shape == km.labels_.shape
            and np.array_equal(np.unique(lab, return_inverse=True)[1], np.unique(km.labels_, return_inverse=True)[1])):
        return km._silhouette[1]
    return engine.pairwise_distances(vals, labels=lab, return_dist=False)[1]
