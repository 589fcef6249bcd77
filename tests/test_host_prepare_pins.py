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


def test_sparse_normalised_counts_round_trip_and_zero_cell_error(tmp_path):
    """Round 5: `prepare_from_matrix` takes the normalised counts as a scipy.sparse matrix -- the reference's sparse branch
    (cnmf.py:537-556: `norm_counts.X` stays sparse and goes to scikit-learn as stored).  The CSR container + label files
    replace the dense frame (never both on disk), the loader hands back the same matrix and labels, the zero-cell error is
    the reference's, and a later dense prepare removes the sparse form."""
    from cnmf_amd.cnmf import SparseFrame
    g = dict(np.load(GOLD, allow_pickle=False))
    cells = ["c%d" % i for i in range(g["norm_counts"].shape[0])]
    genes = list(g["genes"])
    X = sp.csr_matrix(g["norm_counts"])
    obj = cNMF(output_dir=str(tmp_path), name="sp")
    obj.prepare_from_matrix((X, cells, genes), components=[4, 5], n_iter=3, seed=14, beta_loss="kullback-leibler")
    assert os.path.exists(obj.paths["normalized_counts_sparse"]) and not os.path.exists(obj.paths["normalized_counts"])
    assert open(obj.paths["nmf_genes_list"]).read().split("\n") == genes
    fresh = cNMF(output_dir=str(tmp_path), name="sp")           # another process: reads the files
    nc = fresh._load_norm_counts()
    assert isinstance(nc, SparseFrame) and list(nc.index) == cells and list(nc.columns) == genes and nc.shape == X.shape
    assert nc.values.dtype == np.float64 and (nc.values != X).nnz == 0
    bad = X.tolil(); bad[3, :] = 0
    with pytest.raises(Exception, match="Error: 1 cells have zero counts of overdispersed genes. E.g. c3"):
        cNMF(output_dir=str(tmp_path), name="spbad").prepare_from_matrix((bad.tocsr(), cells, genes), components=[4], n_iter=2, seed=1)
    # a dense prepare on the same run directory removes the sparse container (one form on disk, never a stale other one)
    obj.prepare_from_matrix(pd.DataFrame(g["norm_counts"], index=cells, columns=genes), components=[4], n_iter=2, seed=1)
    assert os.path.exists(obj.paths["normalized_counts"]) and not os.path.exists(obj.paths["normalized_counts_sparse"])
    assert isinstance(obj._load_norm_counts(), pd.DataFrame)
