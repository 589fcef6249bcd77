"""The prepare-side artefacts the reference pins in its own reproducibility test
(/root/reference/tests/test_reproducibility.py:118-188: normalized_counts, nmf_replicate_parameters, nmf_run_parameters,
nmf_genes_list, tpm, tpm_stats) as the mirror class writes them from the matrices the UNMODIFIED reference produced
(tests/golden/ref_small.npz, tools/make_golden.py): ledger (n_components, iter, nmf_seed) equal, yaml equal, gene list equal,
normalised counts and TPM statistics within the reference's tolerance (sum of squared differences < 1e-4) -- TPM dense and
sparse.  No device involved (HVG selection / TPM normalisation themselves are out of scope)."""
import os

import numpy as np
import pandas as pd
import pytest
import scipy.sparse as sp
import yaml

from cnmf_amd.cnmf import cNMF, load_df_from_npz

GOLD = os.path.join(os.path.dirname(__file__), "golden", "ref_small.npz")
TOLERANCE = 1e-4


@pytest.mark.parametrize("sparse_tpm", [False, True])
def test_prepare_artefacts_match_the_reference(tmp_path, sparse_tpm):
    g = dict(np.load(GOLD, allow_pickle=False))
    cells = ["c%d" % i for i in range(g["norm_counts"].shape[0])]
    nc = pd.DataFrame(g["norm_counts"], index=cells, columns=list(g["genes"]))
    tpm = pd.DataFrame(g["tpm"], index=cells, columns=list(g["tpm_genes"]))
    obj = cNMF(output_dir=str(tmp_path), name="pins")
    obj.prepare_from_matrix(nc, components=[4, 5, 6], n_iter=12, seed=14, beta_loss="frobenius",
                            tpm=(sp.csr_matrix(g["tpm"]), list(g["tpm_genes"])) if sparse_tpm else tpm)
    # nmf_replicate_parameters: the three columns the reference compares
    led = load_df_from_npz(obj.paths["nmf_replicate_parameters"])
    assert np.array_equal(led[["n_components", "iter", "nmf_seed"]].values.astype(np.int64), g["ledger"])
    assert list(led.columns) == ["n_components", "iter", "nmf_seed", "completed"] and not led["completed"].any()
    # nmf_run_parameters
    kw = yaml.safe_load(open(obj.paths["nmf_run_parameters"]))
    assert kw == dict(alpha_W=0.0, alpha_H=0.0, l1_ratio=0.0, beta_loss="frobenius", solver="cd", tol=1e-4, max_iter=1000,
                      init="random")
    # nmf_genes_list
    assert open(obj.paths["nmf_genes_list"]).read().split("\n") == list(g["genes"])
    # normalized_counts (the reference keeps an h5ad; here the df.npz stand-in): values and labels
    back = load_df_from_npz(obj.paths["normalized_counts"])
    assert list(back.index) == cells and list(back.columns) == list(g["genes"])
    assert ((back.values - g["norm_counts"]) ** 2).sum() < TOLERANCE
    # tpm_stats: __mean / __std per gene as get_mean_var (cnmf.py:126-134, population variance)
    stats = load_df_from_npz(obj.paths["tpm_stats"])
    assert list(stats.columns) == ["__mean", "__std"] and list(stats.index) == list(g["tpm_genes"])
    assert ((stats.values - g["tpm_stats"]) ** 2).sum() < TOLERANCE
    # tpm itself, in whichever container
    if sparse_tpm:
        from cnmf_amd.cnmf import load_csr
        got = np.asarray(load_csr(obj.paths["tpm_sparse"]).todense())
        assert open(obj.paths["tpm_sparse_genes"]).read().split("\n") == list(g["tpm_genes"])
    else:
        got = load_df_from_npz(obj.paths["tpm"]).values
    assert ((got - g["tpm"]) ** 2).sum() < TOLERANCE
