"""Worker for tests/test_dist_gloo.py: launched by torch.distributed.run with world_size 2
on CPU (gloo).  Every rank fabricates the spectra of ITS shard of a ledger deterministically
from (k, iter) -- the engine itself needs a GPU -- and the gather must reproduce, on every
rank, exactly the single-process result in ledger order."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def fake_spectra(k, it, G):
    rs = np.random.RandomState(1000 * k + it)
    return np.abs(rs.standard_normal((k, G))).astype(np.float32)


class GlooEngine:
    """Stand-in for Engine's exchange methods (cnmf_allgather_bytes / cnmf_allgather_spectra
    need a GPU): same signatures and padding contract, transport = gloo.  Lets the packing
    logic of dist.allgather_spectra_rccl run at world_size 2 on CPU."""

    def __init__(self):
        import torch.distributed as dist
        self.comm_world, self.comm_rank = dist.get_world_size(), dist.get_rank()

    def allgather_array(self, a):
        import torch
        import torch.distributed as dist
        a = np.ascontiguousarray(a)
        t = torch.from_numpy(a.view(np.uint8).reshape(-1).copy())
        out = torch.zeros(self.comm_world * t.numel(), dtype=torch.uint8)
        dist.all_gather_into_tensor(out, t)
        return out.numpy().view(a.dtype).reshape((self.comm_world,) + a.shape)

    def allgather_spectra(self, local, rows_max, n_genes=None):
        pad = np.zeros((rows_max, n_genes), dtype=np.float32)
        pad[:local.shape[0]] = local
        return self.allgather_array(pad)


def main():
    import torch.distributed as dist
    from cnmf_amd import dist as cd
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    G = 37
    ledger = [(k, it) for k in (3, 5, 4) for it in range(5)]       # 15 rows, ragged over 2 ranks
    mine = cd.shard(len(ledger), rank, world)
    assert mine == [i for i in range(len(ledger)) if i % world == rank]
    rows = [(i, ledger[i][0], ledger[i][1]) for i in mine]
    hdr, blk = cd.pack_local(rows, [fake_spectra(k, it, G) for _, k, it in rows], G)
    merged = cd.allgather_spectra(hdr, blk, G)
    assert len(merged) == len(ledger)
    for k, it in ledger:
        assert np.array_equal(merged[(k, it)], fake_spectra(k, it, G)), (rank, k, it)
    # an empty shard on one rank must also work
    hdr2, blk2 = cd.pack_local(rows if rank == 0 else [], [fake_spectra(k, it, G) for _, k, it in rows] if rank == 0 else [], G)
    merged2 = cd.allgather_spectra(hdr2, blk2, G)
    assert len(merged2) == len([i for i in range(len(ledger)) if i % world == 0])
    # the library-side packing (transport swapped for gloo) gives the same dictionary
    eng = GlooEngine()
    merged3 = cd.allgather_spectra_rccl(eng, hdr, blk, G)
    assert set(merged3) == set(merged) and all(np.array_equal(merged3[k], merged[k]) for k in merged)
    merged4 = cd.allgather_spectra_rccl(eng, hdr2, blk2, G)
    assert set(merged4) == set(merged2)
    dist.barrier()
    if rank == 0:
        print("DIST_OK world=%d restarts=%d" % (world, len(merged)))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
