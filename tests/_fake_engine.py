"""A device-free stand-in for ``cnmf_amd.engine.Engine`` for tests/test_dist_launcher.py: the PRODUCT's multi-process path
(cnmf_amd/dist.py: launch_ranks -> ``python -m cnmf_amd.dist worker`` -> file rendezvous -> factorize on the rank's shard
-> one all-gather -> rank 0 combines) runs for real in 2 and 4 processes on the CPU; only the device is replaced -- restarts
are fabricated deterministically from their seed, and the collective of the RCCL communicator is played by files in the
launch directory (every rank writes its block, reads everybody's).  Selected with ``engine_factory="tests._fake_engine:make"``;
``CNMF_FAKE_FAIL_RANK`` / ``CNMF_FAKE_HANG_RANK`` make one rank die / never arrive before the communicator forms."""
import os
import time

import numpy as np


def fabricate(k, seed, n_genes):
    rs = np.random.RandomState(int(seed) % (2 ** 31 - 1))
    return np.abs(rs.standard_normal((int(k), int(n_genes)))).astype(np.float32)


class FakeEngine:
    def __init__(self, local_rank):
        self.local_rank = local_rank
        self.shape = None
        self.last_stats = {"outer_iterations": 0}
        self._world, self._rank, self._seq = 1, 0, 0
        self._dir = os.environ.get("CNMF_LAUNCH_DIR")

    # ---- what cNMF.factorize needs
    def set_matrix(self, X):
        self.shape = tuple(np.shape(X))

    def nmf_batch(self, ks, seeds=None, **kw):
        G = self.shape[1]
        H = [fabricate(k, s, G) for k, s in zip(ks, seeds)]
        return H, None, np.full(len(ks), 7, dtype=np.int32), np.zeros(len(ks))

    # ---- the communicator (cnmf_comm_* / cnmf_allgather_*), played by files
    def comm_unique_id(self):
        return bytes([42]) * 128

    def comm_init(self, uid, rank, world):
        assert bytes(uid) == bytes([42]) * 128
        if os.environ.get("CNMF_FAKE_HANG_RANK") == str(rank):
            time.sleep(3600)
        self._rank, self._world = int(rank), int(world)
        self.allgather_array(np.array([rank], dtype=np.int64))          # ncclCommInitRank is collective

    @property
    def comm_world(self):
        return self._world

    @property
    def comm_rank(self):
        return self._rank

    def allgather_array(self, a):
        a = np.ascontiguousarray(a)
        if self._world == 1:
            return a[None].copy()
        seq, self._seq = self._seq, self._seq + 1
        mine = os.path.join(self._dir, "ag_%d_%d.npy" % (seq, self._rank))
        np.save(mine + ".tmp.npy", a)
        os.replace(mine + ".tmp.npy", mine)
        out = []
        for r in range(self._world):
            path = os.path.join(self._dir, "ag_%d_%d.npy" % (seq, r))
            t0 = time.time()
            while not os.path.exists(path):
                if time.time() - t0 > 120:
                    raise TimeoutError("rank %d never wrote %s" % (r, path))
                time.sleep(0.005)
            out.append(np.load(path))
        return np.stack(out)

    def allgather_spectra(self, local, rows_max, n_genes=None):
        pad = np.zeros((int(rows_max), int(n_genes)), dtype=np.float32)
        pad[:local.shape[0]] = local
        return self.allgather_array(pad)


def make(local_rank):
    if os.environ.get("CNMF_FAKE_FAIL_RANK") == str(local_rank):
        raise RuntimeError("no GPU %d on this box (injected by the test)" % local_rank)
    return FakeEngine(local_rank)
