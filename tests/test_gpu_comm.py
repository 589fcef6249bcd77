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
        # round 4: the store outlives a change of matrix (consensus() alternates between the normalised counts and the TPM
        # matrix while the spectra keep serving k selection / further consensus calls) and carries its own gene count;
        # a batch over ANOTHER gene count refuses to append until the store is reset
        eng.set_matrix(X[:100])
        assert eng.spectra_rows == sum(ks) and eng.spectra_genes == X.shape[1]
        assert np.array_equal(eng.spectra_fetch(), ref)
        eng.set_matrix(np.ascontiguousarray(X[:, :300]))
        with pytest.raises(RuntimeError):
            eng.nmf_batch([4], seeds=[3], resident=True)
        eng.spectra_reset()
        eng.nmf_batch([4], seeds=[3], resident=True)
        assert eng.spectra_rows == 4 and eng.spectra_genes == 300


def _bench(args, env, timeout=900):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "CNMF_GATHER", "CNMF_BENCH_BACKEND", "CNMF_RCCL_ID_FILE"):
        e.pop(k, None)
    e.update(env)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=e, capture_output=True, text=True,
                       timeout=timeout, cwd=root)
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    return p, lines, (json.loads(lines[0]) if len(lines) == 1 and p.returncode == 0 else None)


@pytest.mark.parametrize("scaling", ["strong", "weak"])
def test_bench_two_self_spawned_ranks_on_one_gpu(scaling):
    """`python bench.py --gpus 2` SPAWNS its two ranks itself (no torch.distributed.run).  bench.py's N > 1
    bookkeeping (ledger sharding by rank, ragged gather of spectra, barrier, max-over-ranks timing, summed restart
    counts) with two real ranks -- on the one GPU a test box has, over gloo (CNMF_BENCH_BACKEND / CNMF_BENCH_ONE_GPU
    test hooks; real multi-GPU runs use the in-library RCCL gather).  strong: ONE job of 2 x 3 restarts sharded idx % 2;
    weak: 6 restarts per rank."""
    args = ["--gpus", "2", "--steps", "1", "--warmup", "1", "--workload", "C1", "--kmin", "6", "--kmax", "7",
            "--restarts-per-k", "3", "--no-cpu-baseline", "--scaling", scaling]
    p, lines, d = _bench(args, dict(CNMF_BENCH_BACKEND="gloo", CNMF_BENCH_ONE_GPU="1", CNMF_GATHER="torch"))
    assert p.returncode == 0, p.stderr[-3000:]
    assert len(lines) == 1, p.stdout[-2000:]                   # exactly ONE JSON line on stdout
    total = 6 if scaling == "strong" else 12
    assert d["n_gpus"] == 2 and d["steps"] == 1 and d["warmup"] == 1 and d["scaling"] == scaling
    assert d["config"]["restarts_per_step"] == total and d["config"]["restarts_per_step_per_gpu"] == total // 2
    assert d["config"]["gather"] == "torch" and len(d["config"]["per_rank"]) == 2
    assert sum(r["restarts"] for r in d["config"]["per_rank"]) == total
    assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] / 1e3 - total) < 1e-6
    assert "cpu_baseline" not in d and "e2e" not in d          # rank 0 at N = 1 only


def test_bench_gpus_2_on_a_one_gpu_box_fails_loudly():
    """The default transport with more ranks than GPUs must die (rank 1 has no device 1 / RCCL refuses two ranks on
    one device) -- never fall back to a 1-GPU run labelled otherwise."""
    args = ["--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "C1", "--kmin", "6", "--kmax", "6",
            "--restarts-per-k", "2", "--no-cpu-baseline"]
    if Engine(0)._lib.cnmf_device_count() >= 2:
        pytest.skip("this box really has two GPUs")
    p, lines, d = _bench(args, {})
    assert p.returncode != 0 and d is None and not [ln for ln in lines if ln.startswith("{")]
    p, lines, d = _bench(args, dict(CNMF_BENCH_ONE_GPU="1"), timeout=300)       # both on device 0: ncclCommInitRank refuses
    assert p.returncode != 0 and d is None


def test_bench_default_multi_gpu_transport_is_the_library_rccl_gather(tmp_path):
    """bench.py's N > 1 code path with its DEFAULT transport -- the in-library RCCL gather, barrier and
    max-over-ranks, bootstrapped through a file, no torch anywhere -- driven at world = 1 (CNMF_BENCH_FORCE_DIST;
    RCCL refuses two ranks on one device, "Duplicate GPU detected", so a one-GPU box cannot form a larger
    communicator; the N = 2 packing/unpacking is covered over gloo above and in tests/test_dist_gloo.py)."""
    p, lines, d = _bench(["--gpus", "1", "--steps", "1", "--warmup", "1", "--workload", "C1", "--kmin", "6", "--kmax", "7",
                          "--restarts-per-k", "3", "--no-cpu-baseline", "--no-extras"],
                         dict(CNMF_BENCH_FORCE_DIST="1", CNMF_RCCL_ID_FILE=str(tmp_path / "id")))
    assert p.returncode == 0, p.stderr[-3000:]
    # the line itself says what RCCL saw: a communicator of `world` ranks and one all-gather that crossed all of them
    assert d["config"]["rccl"]["communicator_ranks"] == 1 and d["config"]["rccl"]["ranks_seen_by_allgather"] == [0]
    assert len(lines) == 1, p.stdout[-2000:]
    assert d["config"]["gather"] == "rccl" and d["n_gpus"] == 1
    assert d["config"]["torch_in_process"] is False            # the launcher's env is all the N > 1 path needs
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - 6) < 1e-6


def test_bench_comm_dry_run_reports_what_rccl_saw(tmp_path):
    """`bench.py --comm-dry-run`: everything up to the communicator and one all-gather across its ranks, then a JSON line
    that says what RCCL saw -- the readiness check for the first real N > 1 run (here at world 1 through the N > 1 code)."""
    p, lines, d = _bench(["--gpus", "1", "--workload", "C1", "--comm-dry-run"],
                         dict(CNMF_BENCH_FORCE_DIST="1", CNMF_RCCL_ID_FILE=str(tmp_path / "id")), timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1 and d["n_gpus"] == 1 and "dry_run" in d
    assert d["rccl"]["communicator_ranks"] == 1 and d["rccl"]["ranks_seen_by_allgather"] == [0] and d["rccl"]["devices"] == [0]
