"""Generate tests/golden/*.npz by running the UNMODIFIED reference (/root/reference/src/cnmf)
through oracle/scanpy_shim.py in the build container.

The reference's own golden files are download-only (download_pytest_data.py) and there is
no network, so the fixtures are produced here from seeded synthetic counts.  What is stored
is exactly what the reference writes to disk for each stage of the hot path, so the parity
tests can check the new engine stage by stage on the GPU box, where the reference tree does
not exist:

  golden/ref_small.npz
     norm_counts            X handed to the factorisation (prepare output, float64)
     ledger                 rows (n_components, iter, nmf_seed) of nmf_params.df.npz
     merged_k{K}            combined per-restart spectra written by `combine`
     local_density_k{K}     KNN local density cache written by `consensus`
     consensus_spectra_k{K} median spectra (normalised, re-ordered)          cnmf.py:977
     consensus_usages_k{K}  final refit usages                               cnmf.py:978
     stats_k{K}             [k, threshold, silhouette, prediction_error]     cnmf.py:932-936

Run:  python tools/make_golden.py          (takes ~1 min; needs /root/reference)
      python tools/make_golden.py --beta-loss kullback-leibler      -> golden/ref_small_kl.npz
      python tools/make_golden.py --beta-loss itakura-saito         -> golden/ref_small_is.npz   (round 6)
"""
import os
import shutil
import sys
import tempfile
import warnings

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from cnmf_amd import synth  # noqa: E402
from oracle import scanpy_shim  # noqa: E402


def main(beta_loss="frobenius"):
    scanpy_shim.install()
    import cnmf as ref  # the unmodified reference
    from cnmf.cnmf import load_df_from_npz, save_df_to_npz

    # round 5: the same pipeline under beta_loss='kullback-leibler' (solver 'mu' for the restarts AND for the three refits
    # of the consensus tail, cnmf.py:618-631) -> golden/ref_small_kl.npz
    out_path = os.path.join(ROOT, "tests", "golden", {"frobenius": "ref_small.npz", "kullback-leibler": "ref_small_kl.npz",
                                                           "itakura-saito": "ref_small_is.npz"}[beta_loss])
    tmp = tempfile.mkdtemp(prefix="cnmf_golden_")
    try:
        # seeded synthetic counts: 240 cells x 400 genes, 5 programmes
        C, _ = synth.topic_counts(240, 400, 5, mu_lib=7.0, sigma_lib=0.3, seed=7)
        keep = C.sum(axis=0) > 0
        C = C[:, keep]
        if beta_loss == "itakura-saito":
            # scikit-learn REFUSES beta_loss <= 0 on a matrix that contains a zero (sklearn _nmf.py:1679-1684: "the solver
            # may diverge"), i.e. the unmodified reference raises ValueError in factorize() for every ordinary count
            # matrix under --beta-loss itakura-saito.  The only inputs that reach the solver are strictly positive ones:
            # one pseudo-count everywhere.
            C = C + 1
        counts = pd.DataFrame(C.astype(np.int64), index=["c%d" % i for i in range(C.shape[0])],
                              columns=["g%d" % j for j in range(C.shape[1])])
        counts_fn = os.path.join(tmp, "counts.df.npz")
        save_df_to_npz(counts, counts_fn)

        obj = ref.cNMF(output_dir=tmp, name="golden")
        ks = [4, 5, 6]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            obj.prepare(counts_fn, components=ks, n_iter=12, densify=True, seed=14,
                        num_highvar_genes=150, beta_loss=beta_loss)
            obj.factorize(worker_i=0, total_workers=1)
            obj.combine()
            store = {}
            import scanpy as sc
            nc = sc.read(obj.paths["normalized_counts"])
            store["norm_counts"] = np.asarray(nc.X, dtype=np.float64)
            store["genes"] = np.array(list(nc.var.index))
            led = load_df_from_npz(obj.paths["nmf_replicate_parameters"])
            store["ledger"] = led[["n_components", "iter", "nmf_seed"]].values.astype(np.int64)
            for k in ks:
                store["merged_k%d" % k] = load_df_from_npz(obj.paths["merged_spectra"] % k).values
                st = obj.consensus(k, skip_density_and_return_after_stats=True, show_clustering=False)
                store["stats_k%d" % k] = st.values.ravel().astype(np.float64)
            for k, thr in [(5, 0.5), (4, 2.0)]:
                obj.consensus(k, density_threshold=thr, show_clustering=False, build_ref=False)
                rep = str(thr).replace(".", "_")
                store["local_density_k%d" % k] = load_df_from_npz(obj.paths["local_density_cache"] % k).values.ravel()
                store["consensus_spectra_k%d" % k] = load_df_from_npz(obj.paths["consensus_spectra"] % (k, rep)).values
                store["consensus_usages_k%d" % k] = load_df_from_npz(obj.paths["consensus_usages"] % (k, rep)).values
                store["gene_spectra_tpm_k%d" % k] = load_df_from_npz(obj.paths["gene_spectra_tpm"] % (k, rep)).values
                store["gene_spectra_score_k%d" % k] = load_df_from_npz(obj.paths["gene_spectra_score"] % (k, rep)).values
            tpm = sc.read(obj.paths["tpm"])
            store["tpm"] = np.asarray(tpm.X, dtype=np.float64)
            store["tpm_genes"] = np.array(list(tpm.var.index))
            store["tpm_stats"] = load_df_from_npz(obj.paths["tpm_stats"]).values
        os.makedirs(os.path.dirname(out_path), exist_ok=True)
        np.savez_compressed(out_path, **store)
        print("wrote", out_path, {k: np.shape(v) for k, v in store.items()})
        print("size %.1f KiB" % (os.path.getsize(out_path) / 1024))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    bl = "frobenius"
    if "--beta-loss" in sys.argv:
        bl = sys.argv[sys.argv.index("--beta-loss") + 1]
    main(bl)
