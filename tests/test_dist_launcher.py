"""The N-rank launcher of the PRODUCT (cnmf_amd/dist.py: ``cNMF.factorize_multi_gpu`` / ``factorize_multi_process``, the
counterpart of the reference's ``factorize_multi_process``, cnmf.py:677-689) with 2 and 4 real processes on the CPU.
Everything but the device runs as shipped: ``python -m cnmf_amd.dist worker`` per rank, the file rendezvous of the
communicator id, the reference's shard (worker_filter, cnmf.py:52-53), ONE all-gather, rank 0's combine -- the engine is
tests/_fake_engine.py (restarts fabricated from their seeds, the collective played by files).  And the failure modes: a rank
that dies before the communicator forms, a rank that never arrives."""
import os
import time

import numpy as np
import pandas as pd
import pytest

from cnmf_amd import dist as cd
from cnmf_amd.cnmf import cNMF, load_df_from_npz
from tests._fake_engine import fabricate

FACTORY = "tests._fake_engine:make"


def _prepared(tmp_path, n_iter=5, ks=(3, 5, 4), g=17):
    rs = np.random.RandomState(0)
    X = pd.DataFrame(rs.gamma(1.0, 1.0, (40, g)), index=["c%d" % i for i in range(40)], columns=["g%d" % j for j in range(g)])
    obj = cNMF(output_dir=str(tmp_path), name="t")
    obj.prepare_from_matrix(X, components=list(ks), n_iter=n_iter, seed=5)
    return obj, X


def _expected(obj, k, g):
    led = load_df_from_npz(obj.paths["nmf_replicate_parameters"])
    sub = led[led.n_components == k].sort_values("iter")
    return np.concatenate([fabricate(k, s, g) for s in sub["nmf_seed"].values]).astype(np.float64)


@pytest.mark.parametrize("world,gather", [(2, "rccl"), (4, "rccl"), (2, "files")])
def test_product_launcher_shards_gathers_and_combines(tmp_path, world, gather):
    obj, X = _prepared(tmp_path)
    rep = obj.factorize_multi_gpu(n_gpus=world, gather=gather, engine_factory=FACTORY, timeout=240)
    assert rep["world"] == world and rep["gather"] == gather
    if gather == "rccl":
        assert rep["restarts_seen_by_rank0"] == 15                     # every restart of every rank, after ONE gather
    for k in (3, 4, 5):
        merged = load_df_from_npz(obj.paths["merged_spectra"] % k)
        assert list(merged.index) == ["iter%d_topic%d" % (it, t) for it in range(5) for t in range(1, k + 1)]
        assert list(merged.columns) == list(X.columns)
        assert np.allclose(merged.values, _expected(obj, k, X.shape[1]), rtol=0, atol=0)
    # the reference's per-iteration files exist too (the resume contract: cnmf.py:742-745, 729-733)
    assert os.path.exists(obj.paths["iter_spectra"] % (3, 0)) and os.path.exists(obj.paths["iter_spectra"] % (5, 4))


def test_reference_entry_point_factorize_multi_process_needs_the_gpus(tmp_path):
    obj, _ = _prepared(tmp_path)
    with pytest.raises((ValueError, ImportError, OSError, RuntimeError)):
        obj.factorize_multi_process(64)                                # more workers than GPUs (or no library at all here)


def test_rank_dying_before_the_communicator_forms_is_a_clear_error(tmp_path, monkeypatch):
    obj, _ = _prepared(tmp_path)
    monkeypatch.setenv("CNMF_FAKE_FAIL_RANK", "2")
    t0 = time.time()
    with pytest.raises(cd.RankFailure) as ei:
        obj.factorize_multi_gpu(n_gpus=4, engine_factory=FACTORY, timeout=240)
    msg = str(ei.value)
    assert time.time() - t0 < 120                                      # stopped at once, not at the timeout
    assert "rank 2 of 4 exited with code" in msg and "'imported'" in msg and "no GPU 2 on this box" in msg
    assert "rank 0:" in msg and "rank 3:" in msg                       # where the survivors were when they were killed
    assert not os.path.exists(obj.paths["merged_spectra"] % 3)         # and nothing half-written is left as a result


def test_rank_that_never_arrives_times_out_with_the_stages(tmp_path, monkeypatch):
    obj, _ = _prepared(tmp_path)
    monkeypatch.setenv("CNMF_FAKE_HANG_RANK", "1")
    with pytest.raises(cd.RankFailure) as ei:
        obj.factorize_multi_gpu(n_gpus=2, engine_factory=FACTORY, timeout=20)
    msg = str(ei.value)
    assert "no result after 20 s" in msg and "rank 1: 'rendezvous'" in msg
