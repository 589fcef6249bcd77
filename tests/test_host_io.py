"""Host-side file helpers of the mirror class (cnmf_amd/cnmf.py): the npz containers of cnmf.py:31-40 and the
large-matrix variant that keeps the data beside the npz."""
import numpy as np
import pandas as pd

from cnmf_amd import cnmf as m


def _frame(n, g, seed=0):
    rng = np.random.RandomState(seed)
    return pd.DataFrame(rng.gamma(1.0, 1.0, (n, g)), index=["c%d" % i for i in range(n)], columns=["g%d" % j for j in range(g)])


def test_npz_round_trip_small_and_with_sibling(tmp_path, monkeypatch):
    df = _frame(40, 7)
    for name, limit in (("small.df.npz", 1 << 40), ("big.df.npz", 1)):
        monkeypatch.setattr(m, "_SIBLING_BYTES", limit)
        path = str(tmp_path / name)
        m.save_df_to_npz_fast(df, path, sibling_ok=True)
        got = m.load_df_from_npz(path)
        assert np.array_equal(got.values, df.values)
        assert list(got.index) == list(df.index) and list(got.columns) == list(df.columns)
    assert (tmp_path / "big.df.npz.data.npy").exists() and not (tmp_path / "small.df.npz.data.npy").exists()
    with np.load(str(tmp_path / "small.df.npz"), allow_pickle=True) as f:          # the reference's own container (cnmf.py:31-32)
        assert sorted(f.files) == ["columns", "data", "index"]
    m.save_df_to_npz_fast(df, str(tmp_path / "merged.df.npz"))                     # merged spectra: always the reference's form
    with np.load(str(tmp_path / "merged.df.npz"), allow_pickle=True) as f:
        assert sorted(f.files) == ["columns", "data", "index"]


def test_compressed_container_is_the_reference_one(tmp_path):
    df = _frame(5, 3, seed=1)
    path = str(tmp_path / "x.df.npz")
    m.save_df_to_npz(df, path)
    with np.load(path, allow_pickle=True) as f:
        assert sorted(f.files) == ["columns", "data", "index"]
        assert np.array_equal(f["data"], df.values)


def test_sparse_tpm_statistics_match_get_mean_var(tmp_path):
    """prepare_from_matrix with a sparse TPM matrix: __mean / __std of tpm_stats as the reference's get_mean_var computes
    them for sparse input (cnmf.py:126-134: mean, E[x^2] - mean^2, population), duplicates in the input summed first."""
    import scipy.sparse as sp
    rs = np.random.RandomState(2)
    dense = rs.gamma(0.5, 2.0, (60, 9)) * (rs.uniform(size=(60, 9)) < 0.4)
    coo = sp.coo_matrix(dense)
    dup = sp.coo_matrix((np.concatenate([coo.data * 0.25, coo.data * 0.75]), (np.concatenate([coo.row, coo.row]), np.concatenate([coo.col, coo.col]))), shape=dense.shape)
    X = _frame(60, 5)
    obj = m.cNMF(output_dir=str(tmp_path), name="s")
    obj.prepare_from_matrix(X, components=[3], n_iter=2, seed=1, tpm=(dup, ["t%d" % j for j in range(9)]))
    stats = m.load_df_from_npz(obj.paths["tpm_stats"])
    mean = dense.mean(axis=0)
    std = np.sqrt((dense * dense).mean(axis=0) - mean ** 2)
    assert np.allclose(stats["__mean"].values, mean, rtol=1e-12, atol=0)
    assert np.allclose(stats["__std"].values, std, rtol=1e-10, atol=0)
    assert list(stats.index) == ["t%d" % j for j in range(9)]


def test_text_writer_is_byte_identical_to_pandas(tmp_path):
    """save_df_to_text (cnmf.py:34-35) formats all-float64 frames itself: the bytes must be those of
    ``DataFrame.to_csv(sep='\\t')`` -- values across the whole double range, integer and string labels -- and every
    frame it does not take (float32, NaN, labels with a tab or a quote, named index) must go through pandas unchanged."""
    rs = np.random.RandomState(4)
    v = rs.gamma(0.3, 1.0, (400, 7)) * np.exp(9 * rs.standard_normal((400, 7)))
    v[0, 0], v[1, 1], v[2, 2], v[3, 3], v[4, 4], v[5, 5], v[6, 6] = 0.0, 5e-324, 1e22, 123456789.0, 1e16, 0.1, -2.5e-7
    frames = {
        "usages": pd.DataFrame(v, index=["cell_%d" % i for i in range(400)], columns=np.arange(1, 8)),
        "spectra": pd.DataFrame(v.T.copy(), index=np.arange(1, 8), columns=["g%d" % j for j in range(400)]),
        "f32": pd.DataFrame(v.astype(np.float32), index=["c%d" % i for i in range(400)], columns=np.arange(1, 8)),
        "nan": pd.DataFrame(np.where(v > 100, np.nan, v), index=["c%d" % i for i in range(400)], columns=np.arange(1, 8)),
        "tab": pd.DataFrame(v[:3], index=["a\tb", 'q"x', "plain"], columns=np.arange(1, 8)),
        "named": pd.DataFrame(v[:3], index=pd.Index(["a", "b", "c"], name="cell"), columns=np.arange(1, 8)),
        # labels whose str() is NOT what pandas writes (round-3 advisor): None / NaN -> '', timestamps drop midnight
        "none_label": pd.DataFrame(v[:3], index=["a", None, "c"], columns=np.arange(1, 8)),
        "nan_label": pd.DataFrame(v[:3], index=pd.Index(["a", np.nan, "c"], dtype=object), columns=np.arange(1, 8)),
        "dates": pd.DataFrame(v[:3], index=pd.date_range("2020-01-01", periods=3), columns=np.arange(1, 8)),
        "float_labels": pd.DataFrame(v[:3], index=[0.5, 1.0, 2.0], columns=np.arange(1, 8)),
        "series": pd.Series(v[:5, 0], index=["a", "b", "c", "d", "e"]),
    }
    for name, df in frames.items():
        got, ref = str(tmp_path / (name + ".got.txt")), str(tmp_path / (name + ".ref.txt"))
        m.save_df_to_text(df, got)
        df.to_csv(ref, sep="\t")
        assert open(got, "rb").read() == open(ref, "rb").read(), name


def test_large_float_tables_are_stored_not_deflated_and_read_back_identically(tmp_path):
    """save_df_to_npz keeps the reference's container (cnmf.py:31-32: three members data / index / columns read back with
    np.load) -- a float ``data`` member of a megabyte or more is stored instead of deflated (zlib gains ~1 % on float64
    mantissas and was the critical path of consensus()'s artefact writes); small tables are deflated as before."""
    import zipfile
    from cnmf_amd.cnmf import save_df_to_npz, load_df_from_npz
    rs = np.random.RandomState(0)
    big = pd.DataFrame(rs.rand(20000, 9), index=["cell%d" % i for i in range(20000)], columns=np.arange(1, 10))
    small = pd.DataFrame(rs.rand(9, 300), index=np.arange(1, 10), columns=["g%d" % j for j in range(300)])
    for name, df, stored in (("big", big, True), ("small", small, False)):
        fn = str(tmp_path / (name + ".df.npz"))
        save_df_to_npz(df, fn)
        with zipfile.ZipFile(fn) as zf:
            info = {i.filename: i.compress_type for i in zf.infolist()}
        assert set(info) == {"data.npy", "index.npy", "columns.npy"}
        assert (info["data.npy"] == zipfile.ZIP_STORED) == stored and info["index.npy"] == zipfile.ZIP_DEFLATED
        with np.load(fn, allow_pickle=True) as f:                  # exactly the reference's load_df_from_npz (cnmf.py:36-38)
            back = pd.DataFrame(**f)
        assert back.equals(df) and load_df_from_npz(fn).equals(df)
