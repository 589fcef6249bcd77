"""tests/golden/ref_hvg.npz: the reference's high-variance-gene statistics (cnmf.py:192-246, dense branch, and
:136-188, sparse branch) on a seeded synthetic TPM matrix, produced by the UNMODIFIED reference through
oracle/scanpy_shim.py in the build container (no network, no reference tree on the GPU box).

Run:  python tools/make_golden_hvg.py"""
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from cnmf_amd import synth  # noqa: E402
from oracle import scanpy_shim  # noqa: E402


def main():
    scanpy_shim.install()
    from cnmf.cnmf import get_highvar_genes, get_highvar_genes_sparse

    C, _ = synth.topic_counts(300, 600, 6, mu_lib=7.0, sigma_lib=0.4, seed=11)
    C = C[:, C.sum(axis=0) > 0]
    C = C[C.sum(axis=1) > 0].astype(np.float64)
    tpm = C / C.sum(axis=1, keepdims=True) * 1e6
    store = {"tpm": tpm}
    for tag, kw in (("n200", dict(numgenes=200)), ("thr", dict())):
        stats, params = get_highvar_genes(tpm, **kw)
        for col in ("mean", "var", "fano", "expected_fano", "fano_ratio"):
            store["%s_%s" % (tag, col)] = stats[col].values.astype(np.float64)
        store["%s_high_var" % tag] = np.asarray(stats["high_var"].values, dtype=bool)
        store["%s_params" % tag] = np.array([params["A"], params["B"], np.nan if params["T"] is None else params["T"],
                                             params["minimal_mean"]], dtype=np.float64)
    stats, params = get_highvar_genes_sparse(sp.csr_matrix(tpm), numgenes=200)
    store["sparse_n200_high_var"] = np.asarray(stats["high_var"].values, dtype=bool)
    store["sparse_n200_var"] = stats["var"].values.astype(np.float64)
    out = os.path.join(ROOT, "tests", "golden", "ref_hvg.npz")
    np.savez_compressed(out, **store)
    print("wrote", out, {k: v.shape for k, v in store.items()})


if __name__ == "__main__":
    main()
