"""High-variance-gene statistics against the UNMODIFIED reference (tests/golden/ref_hvg.npz, written by
tools/make_golden_hvg.py): the O(genes) restatement on the CPU, the device moments on the GPU."""
import os

import numpy as np
import pytest

from cnmf_amd.hvg import highvar_genes_from_moments

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_hvg.npz"))


@pytest.mark.parametrize("tag,kw", [("n200", dict(numgenes=200)), ("thr", dict())])
def test_hvg_statistics_match_reference(tag, kw):
    tpm = G["tpm"]
    stats, params = highvar_genes_from_moments(tpm.mean(axis=0), tpm.var(axis=0, ddof=0), **kw)
    for col in ("mean", "var", "fano", "expected_fano", "fano_ratio"):
        assert np.allclose(stats[col].values, G["%s_%s" % (tag, col)], rtol=1e-12, atol=0), col
    assert np.array_equal(stats["high_var"].values, G["%s_high_var" % tag])
    ref = G["%s_params" % tag]
    assert np.isclose(params["A"], ref[0], rtol=1e-12) and np.isclose(params["B"], ref[1], rtol=1e-12)
    assert (params["T"] is None and np.isnan(ref[2])) or np.isclose(params["T"], ref[2], rtol=1e-12)
    if tag == "n200":
        assert stats["high_var"].sum() == 200
        assert np.array_equal(stats["high_var"].values, G["sparse_n200_high_var"])     # the sparse branch agrees


@pytest.mark.gpu
def test_hvg_selection_with_device_moments(engine, tmp_path):
    import scipy.sparse as sp
    from cnmf_amd.cnmf import cNMF
    tpm = G["tpm"]
    for X in (tpm, sp.csr_matrix(tpm.astype(np.float32))):
        engine.set_matrix(X)
        mean, var = engine.col_mean_var()
        # the resident matrix is float32: moments of the rounded values, accumulated in float64
        assert np.allclose(mean, tpm.mean(axis=0), rtol=1e-6) and np.allclose(var, tpm.var(axis=0), rtol=1e-5)
        stats, _ = highvar_genes_from_moments(mean, var, numgenes=200)
        assert (stats["high_var"].values != G["n200_high_var"]).sum() <= 2        # ties at the 200th rank only
    obj = cNMF(output_dir=str(tmp_path), name="hvg", engine=engine)
    stats, params, chosen = obj.select_highvar_genes(tpm, numgenes=200)
    assert len(chosen) == 200 and abs(params["A"] - G["n200_params"][0]) <= 1e-6 * G["n200_params"][0]
