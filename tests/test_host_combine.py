"""combine_nmf of the mirror class (cnmf.py:748-773) without a device: restarts kept in memory by factorize (plain
arrays + the gene names) and restarts read from per-iteration files gather into the same merged frame -- labels
``iter%d_topic%d`` in (iter, topic) order, ENOENT for a missing restart unless skip_missing_files."""
import errno

import numpy as np
import pandas as pd
import pytest

from cnmf_amd.cnmf import cNMF, load_df_from_npz, save_df_to_npz


def _prepared(tmp_path, n_iter=3, k=4, g=11):
    rs = np.random.RandomState(0)
    X = pd.DataFrame(rs.gamma(1.0, 1.0, (30, g)), index=["c%d" % i for i in range(30)], columns=["g%d" % j for j in range(g)])
    obj = cNMF(output_dir=str(tmp_path), name="t")
    obj.prepare_from_matrix(X, components=[k], n_iter=n_iter, seed=5)
    spectra = {it: rs.gamma(1.0, 1.0, (k, g)) for it in range(n_iter)}
    return obj, X, spectra


def test_memory_and_file_restarts_gather_alike(tmp_path, capsys):
    obj, X, spectra = _prepared(tmp_path)
    k = 4
    # restart 0 and 2 in memory (as factorize leaves them), restart 1 only as the reference's per-iteration file
    obj._spectra_columns = X.columns
    obj.spectra_cache[(k, 0)] = spectra[0]
    obj.spectra_cache[(k, 2)] = spectra[2]
    save_df_to_npz(pd.DataFrame(spectra[1], index=np.arange(1, k + 1), columns=X.columns), obj.paths["iter_spectra"] % (k, 1))
    merged = obj.combine_nmf(k)
    assert list(merged.index) == ["iter%d_topic%d" % (it, t) for it in range(3) for t in range(1, k + 1)]
    assert list(merged.columns) == list(X.columns)
    assert np.array_equal(merged.values, np.concatenate([spectra[0], spectra[1], spectra[2]]))
    on_disk = load_df_from_npz(obj.paths["merged_spectra"] % k)
    assert np.array_equal(on_disk.values, merged.values) and list(on_disk.index) == list(merged.index)
    assert "Combining factorizations for k=4." in capsys.readouterr().out


def test_missing_restart_raises_enoent_or_is_skipped(tmp_path):
    obj, X, spectra = _prepared(tmp_path)
    k = 4
    obj._spectra_columns = X.columns
    obj.spectra_cache[(k, 0)] = spectra[0]
    with pytest.raises(FileNotFoundError) as ei:
        obj.combine_nmf(k)
    assert ei.value.errno == errno.ENOENT
    merged = obj.combine_nmf(k, skip_missing_files=True)
    assert list(merged.index) == ["iter0_topic%d" % t for t in range(1, k + 1)]
    assert np.array_equal(merged.values, spectra[0])
