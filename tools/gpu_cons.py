"""GPU probe: consensus-only stress (BASELINE config 5) wall-clock, GPU vs sklearn/pandas on the host."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cnmf_amd import synth
from cnmf_amd.engine import Engine
eng = Engine(0)
S, truth = synth.consensus_stress(R=5000, G=2000, k=20, n_outliers=100, seed=0)
for rep in range(3):
    t = time.perf_counter(); out = eng.consensus(S, 20, density_threshold=0.5); dt = time.perf_counter() - t
    print("GPU consensus core (5000x2000,k=20): %.1f ms  kept %d  kmeans_iter %d" % (dt * 1e3, out["n_kept"], out["kmeans_n_iter"]), flush=True)
t = time.perf_counter(); out = eng.consensus(S, 20, skip_density=True, want_silhouette=True); dt = time.perf_counter() - t
print("GPU stats-mode (kmeans on all rows + silhouette): %.1f ms" % (dt * 1e3), flush=True)
if os.environ.get("CPU", "1") == "1":
    from oracle import consensus as oc
    from sklearn.cluster import KMeans
    from sklearn.metrics.pairwise import euclidean_distances
    import pandas as pd
    t0 = time.perf_counter()
    l2 = oc.l2_normalise(S); t1 = time.perf_counter()
    D = euclidean_distances(l2); t2 = time.perf_counter()
    n = int(0.3 * 5000 / 20)
    po = np.argpartition(D, n + 1)[:, :n + 1]; dens = D[np.arange(5000)[:, None], po].sum(1) / n; t3 = time.perf_counter()
    keep = dens < 0.5; l2k = l2[keep]
    km = KMeans(n_clusters=20, n_init=10, random_state=1).fit(l2k); t4 = time.perf_counter()
    med = pd.DataFrame(l2k).groupby(pd.Series(km.labels_ + 1)).median(); t5 = time.perf_counter()
    print("CPU (cores %d): l2 %.2f s, dist %.2f s, knn %.2f s, kmeans %.2f s, median %.2f s, total %.2f s"
          % (os.cpu_count(), t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t5 - t0), flush=True)
    kept = out["density_filter"] if "density_filter" in out else None
