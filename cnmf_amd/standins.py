"""Device-backed stand-ins for the three scikit-learn names the reference's ``consensus`` body calls
(cnmf.py:15-18: ``KMeans`` :908, ``euclidean_distances`` :891/988, ``silhouette_score`` :923).

``integration/hip_backend.py`` (INTEGRATION.md Option B: a subclass of the UNMODIFIED reference class) swaps them in
for the duration of ``consensus``.  They live in the package -- not next to the subclass -- because they need nothing
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
    """``euclidean_distances(l2_spectra)`` (cnmf.py:891, 988).  The rows are already L2-normalised (cnmf.py:882): the
    distance matrix of the device's consensus core (which normalises again: idempotent to the last ulp)."""
    if Y is not None:
        raise NotImplementedError("the device stand-in computes all-pairs distances of ONE matrix (cnmf.py:891)")
    vals = X.values if hasattr(X, "values") else np.asarray(X)
    return engine.consensus(vals, 1, skip_density=True, return_dist=True, n_init=1)["topics_dist"]


def device_silhouette_score(last_kmeans, fallback, X, labels, metric="euclidean", **kw):
    """``silhouette_score(l2_spectra, labels, metric='euclidean')`` (cnmf.py:923): the fit of cnmf.py:909 already
    produced it on the device for exactly these rows; anything else goes to ``fallback`` (the real function)."""
    km = last_kmeans
    if km is not None and km._silhouette[0] == np.shape(X) and metric == "euclidean":
        return km._silhouette[1]
    return fallback(X, labels, metric=metric, **kw)
