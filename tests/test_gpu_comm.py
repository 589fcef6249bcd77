"""The library's own exchange (include/cnmf_hip.h "multi-GPU exchange"): RCCL bound lazily with
dlopen, a real communicator (world = 1 is all a 1-GPU box offers; the N > 1 packing logic is
covered on CPU by tests/test_dist_gloo.py), the resident spectra store, ragged zero-padding."""
import numpy as np
import pytest

from cnmf_amd import dist, synth
from cnmf_amd.engine import Engine

pytestmark = pytest.mark.gpu


def test_gather_without_communicator_is_a_copy(engine):
    X = synth.make_config("C1", dtype=np.float32, n_cells=300)
    engine.set_matrix(X)
    assert engine.comm_world == 1 and engine.comm_rank == 0
    blk = np.random.RandomState(0).rand(17, X.shape[1]).astype(np.float32)
    out = engine.allgather_spectra(blk, rows_max=20)
    assert out.shape == (1, 20, X.shape[1])
    assert np.array_equal(out[0, :17], blk) and not out[0, 17:].any()      # ragged shard zero-padded
    a = np.arange(12, dtype=np.int64).reshape(3, 4)
    assert np.array_equal(engine.allgather_array(a)[0], a)


def test_rccl_communicator_world1_and_resident_store(tmp_path):
    """ncclGetUniqueId / ncclCommInitRank / ncclAllGather through the C-ABI, spectra taken
    straight from the device-resident store (they never visit the host before the exchange)."""
    X = synth.make_config("C1", dtype=np.float32, n_cells=400)
    with Engine(0) as eng:
        eng.set_matrix(X)
        dist.comm_bootstrap_file(eng, 0, 1, str(tmp_path / "rccl_id"))
        assert eng.comm_world == 1
        with pytest.raises(RuntimeError):
            eng.comm_init(eng.comm_unique_id(), 0, 1)                      # already initialised
        ks, seeds = [5, 7, 6], [11, 12, 13]
        H, _, n_iter, _ = eng.nmf_batch(ks, seeds=seeds)
        eng.spectra_reset()
        eng.nmf_batch(ks[:2], seeds=seeds[:2], resident=True)
        eng.nmf_batch(ks[2:], seeds=seeds[2:], resident=True)              # appends
        assert eng.spectra_rows == sum(ks)
        ref = np.concatenate(H, axis=0)
        assert np.array_equal(eng.spectra_fetch(), ref)                    # same kernels, same bits
        hdr = np.array([(i, k, i) for i, k in enumerate(ks)], dtype=np.int32)
        merged = dist.allgather_spectra_rccl(eng, hdr, None, X.shape[1])   # blk=None -> resident store
        assert set(merged) == {(k, i) for i, k in enumerate(ks)}
        for i, k in enumerate(ks):
            assert np.array_equal(merged[(k, i)], H[i])
        # host-block route gives the same answer
        merged2 = dist.allgather_spectra_rccl(eng, hdr, ref, X.shape[1])
        for key in merged:
            assert np.array_equal(merged[key], merged2[key])
        with pytest.raises(ValueError):
            eng.allgather_spectra(None, rows_max=3)                        # rows_max < rows held
        eng.comm_finalize()
        assert eng.comm_world == 1
        # a new matrix invalidates the store
        eng.set_matrix(X[:100])
        assert eng.spectra_rows == 0


def test_bench_two_ranks_on_one_gpu():
    """bench.py's N > 1 bookkeeping (ledger sharding by rank, ragged gather of spectra, barrier, max-over-ranks
    timing, summed restart counts) with two real ranks -- on the one GPU a test box has, over gloo
    (CNMF_BENCH_BACKEND / CNMF_BENCH_ONE_GPU test hooks; the driver's multi-GPU runs use nccl = RCCL)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CNMF_BENCH_BACKEND="gloo", CNMF_BENCH_ONE_GPU="1", CNMF_GATHER="torch",
               MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29561", os.path.join(root, "bench.py"),
           "--gpus", "2", "--steps", "1", "--warmup", "1", "--workload", "C1", "--kmin", "6", "--kmax", "7",
           "--restarts-per-k", "3", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[-2000:]                   # exactly ONE JSON line on stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 1 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["restarts_per_step_per_gpu"] == 6 and d["config"]["gather"] == "torch"
    assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] / 1e3 - 12) < 1e-6      # 2 ranks x 6 restarts
    assert "cpu_baseline" not in d                             # rank 0 at N = 1 only


def test_bench_default_multi_gpu_transport_is_the_library_rccl_gather(tmp_path):
    """bench.py's N > 1 code path with its DEFAULT transport -- the in-library RCCL gather, barrier and
    max-over-ranks, bootstrapped through a file, no torch anywhere -- driven at world = 1 (CNMF_BENCH_FORCE_DIST;
    RCCL refuses two ranks on one device, "Duplicate GPU detected", so a one-GPU box cannot form a larger
    communicator; the N = 2 packing/unpacking is covered over gloo above and in tests/test_dist_gloo.py)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CNMF_BENCH_FORCE_DIST="1", CNMF_RCCL_ID_FILE=str(tmp_path / "id"))
    env.pop("CNMF_GATHER", None); env.pop("CNMF_BENCH_BACKEND", None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1",
           "--workload", "C1", "--kmin", "6", "--kmax", "7", "--restarts-per-k", "3", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["config"]["gather"] == "rccl" and d["n_gpus"] == 1
    assert d["config"]["torch_in_process"] is False            # the launcher's env is all the N > 1 path needs
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - 6) < 1e-6
