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
