"""world_size-2 test of the restart sharding + the single all-gather of spectra on CPU
(gloo backend), as the driver cannot give this round more than one GPU.  The same code path
runs with backend nccl (= RCCL) and GPU tensors in bench.py --gpus N."""
import os
import subprocess
import sys

import numpy as np

from cnmf_amd import dist as cd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_is_reference_worker_filter():
    for world in (1, 2, 3, 8):
        seen = []
        for rank in range(world):
            mine = cd.shard(23, rank, world)
            assert mine == [p for i, p in enumerate(range(23)) if (i - rank) % world == 0]   # cnmf.py:52-53
            seen += mine
        assert sorted(seen) == list(range(23))


def test_pack_unpack_roundtrip():
    rs = np.random.RandomState(0)
    rows = [(0, 3, 0), (2, 5, 1), (4, 4, 7)]
    spectra = [rs.rand(k, 11).astype(np.float32) for _, k, _ in rows]
    hdr, blk = cd.pack_local(rows, spectra, 11)
    assert blk.shape == (12, 11)
    out = cd.unpack([hdr], [blk])
    for (_, k, it), s in zip(rows, spectra):
        assert np.array_equal(out[(k, it)], s)
    # single process, no process group: allgather degenerates to unpack
    out2 = cd.allgather_spectra(hdr, blk, 11)
    assert set(out2) == set(out)


def test_allgather_two_ranks_gloo():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29531",
           os.path.join(ROOT, "tests", "_dist_worker.py")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "DIST_OK world=2 restarts=15" in p.stdout
