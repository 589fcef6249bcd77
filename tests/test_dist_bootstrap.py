"""The torch-free rendezvous of the in-library RCCL communicator (cnmf_amd/dist.py::comm_bootstrap_file, used by
`bench.py --gpus N` and `factorize_distributed(gather="rccl")`), exercised with several real processes on the CPU:
rank 0 publishes the 128-byte id through a file, every rank must call comm_init with the SAME id and its own rank,
the id file must be gone afterwards, and a stale file from an earlier launch must not be picked up by rank 0's peers
once rank 0 has replaced it.  (The collective itself needs GPUs; here the engine is a recorder.)"""
import multiprocessing as mp
import os
import time

import pytest

from cnmf_amd import dist as cd


class _RecorderEngine:
    def __init__(self, rank, barrier):
        self.rank, self.barrier, self.calls = rank, barrier, []

    def comm_unique_id(self):
        return bytes([17 + self.rank]) * 128          # only rank 0's id may ever be used

    def comm_init(self, uid, rank, world):
        self.calls.append((bytes(uid), rank, world))
        self.barrier.wait(timeout=30)                  # ncclCommInitRank is collective: returns when all ranks are in


def _worker(rank, world, path, barrier, q, delay):
    time.sleep(delay)
    eng = _RecorderEngine(rank, barrier)
    cd.comm_bootstrap_file(eng, rank, world, path, timeout=30.0)
    q.put((rank, eng.calls))


@pytest.mark.parametrize("world,rank0_late", [(2, False), (4, True)])
def test_file_rendezvous_same_id_on_every_rank(tmp_path, world, rank0_late):
    ctx = mp.get_context("fork")
    path = str(tmp_path / "rccl_id")
    barrier, q = ctx.Barrier(world), ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, path, barrier, q, 0.5 if (r == 0 and rank0_late) else 0.0))
             for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=60) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for r in range(world):
        assert got[r] == [(bytes([17]) * 128, r, world)], (r, got[r])
    assert not os.path.exists(path)


def test_world_one_needs_no_file(tmp_path):
    class E:
        def comm_unique_id(self): return b"\x01" * 128
        def comm_init(self, uid, rank, world): self.got = (uid, rank, world)
    e = E()
    cd.comm_bootstrap_file(e, 0, 1, str(tmp_path / "never_written"))
    assert e.got == (b"\x01" * 128, 0, 1) and not os.path.exists(str(tmp_path / "never_written"))
