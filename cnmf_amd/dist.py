"""Multi-GPU sharding of the restart ledger + the single gather of spectra.

The restart hot loop shards with no data-path collective: ledger row ``idx`` goes to rank
``idx % world`` -- exactly the reference's ``worker_filter`` (cnmf.py:52-53), so the
``completed`` / resume semantics carry over.  Every rank holds a replica of X (400 MB at
50k x 2000, 1.6 GB at 200k x 2000, versus 288 GB of HBM).  The reference's "gather" is the
filesystem (``combine`` re-reads one npz per restart, cnmf.py:755-770); here it is ONE
``all_gather`` of the packed float32 spectra (<= 130 MB in total for the largest BASELINE
config, i.e. latency- not bandwidth-bound on xGMI).  Two transports, same packing:

* ``allgather_spectra_rccl`` -- ``ncclAllGather`` inside the C-ABI library
  (``cnmf_allgather_spectra``, include/cnmf_hip.h): no torch anywhere; the 128-byte RCCL id
  travels through a file (``comm_bootstrap_file``) or any launcher's own store.
* ``allgather_spectra`` -- ``torch.distributed`` (backend "nccl" IS RCCL on ROCm; "gloo" for
  the CPU tests); torch is plumbing there (process group + collective), nothing numeric.

After the gather every k's consensus is independent again.
"""
import numpy as np


def shard(n_rows, rank, world):
    """Ledger rows owned by ``rank``: (i - rank) % world == 0  (cnmf.py:52-53)."""
    return [i for i in range(n_rows) if (i - rank) % world == 0]


def pack_local(ledger_rows, spectra_list, n_genes):
    """Pack this rank's spectra into one float32 block + an int32 header per restart.

    ledger_rows : list of (ledger_index, k, iter)   spectra_list : list of (k x G) arrays
    Returns (header [n,3] int32, block [sum k, G] float32)."""
    hdr = np.asarray(ledger_rows, dtype=np.int32).reshape(-1, 3)
    if len(spectra_list):
        blk = np.ascontiguousarray(np.concatenate([np.asarray(s, dtype=np.float32) for s in spectra_list], axis=0))
    else:
        blk = np.zeros((0, n_genes), dtype=np.float32)
    assert blk.shape[0] == int(hdr[:, 1].sum()) if len(hdr) else blk.shape[0] == 0
    return hdr, blk


def allgather_spectra(hdr, blk, n_genes, device=None):
    """One padded all-gather of the packed spectra (plus one tiny one for the headers).

    Returns {(k, iter): spectra ndarray} for ALL restarts of ALL ranks, on every rank."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return unpack([hdr], [blk])
    world = dist.get_world_size()
    dev = torch.device(device) if device is not None else torch.device("cpu")
    # sizes first (restart count, row count) so that ragged shards can be padded
    mine = torch.tensor([hdr.shape[0], blk.shape[0]], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, mine)
    sizes = [tuple(int(v) for v in s.tolist()) for s in sizes]
    max_n = max(1, max(s[0] for s in sizes))
    max_rows = max(1, max(s[1] for s in sizes))
    h = torch.zeros((max_n, 3), dtype=torch.int32, device=dev)
    h[:hdr.shape[0]] = torch.from_numpy(hdr).to(dev)
    b = torch.zeros((max_rows, n_genes), dtype=torch.float32, device=dev)
    if blk.shape[0]:
        b[:blk.shape[0]] = torch.from_numpy(blk).to(dev)
    hs = torch.zeros(world * max_n * 3, dtype=torch.int32, device=dev)
    bs = torch.zeros(world * max_rows * n_genes, dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(hs, h.reshape(-1))
    dist.all_gather_into_tensor(bs, b.reshape(-1))   # THE data-path collective (RCCL over xGMI on GPUs)
    hs = hs.cpu().numpy().reshape(world, max_n, 3)
    bs = bs.cpu().numpy().reshape(world, max_rows, n_genes)
    return unpack([hs[r, :sizes[r][0]] for r in range(world)], [bs[r, :sizes[r][1]] for r in range(world)])


def comm_bootstrap_file(engine, rank, world, path, timeout=300.0):
    """Torch-free rendezvous for the library's RCCL communicator: rank 0 writes the id to
    ``path`` (atomically), the others wait for it; then the collective ``comm_init``."""
    import os
    import time
    if world == 1:
        return engine.comm_init(engine.comm_unique_id(), 0, 1)
    if rank == 0:
        uid = engine.comm_unique_id()
        tmp = "%s.tmp.%d" % (path, os.getpid())
        with open(tmp, "wb") as f:
            f.write(uid)
        os.replace(tmp, path)
    else:
        t0 = time.time()
        while not os.path.exists(path):
            if time.time() - t0 > timeout:
                raise TimeoutError("no RCCL id at %s after %.0f s" % (path, timeout))
            time.sleep(0.05)
        with open(path, "rb") as f:
            uid = f.read()
    engine.comm_init(uid, rank, world)          # collective: once it returns every rank has read the file
    if rank == 0:
        try:
            os.remove(path)                      # never leave a stale id behind for a later launch
        except OSError:
            pass


def comm_bootstrap_torch(engine):
    """Same, when a torch.distributed process group already exists (bench.py's launcher):
    the id is broadcast through it; the data path stays inside the library."""
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    box = [engine.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    engine.comm_init(box[0], rank, world)


def allgather_spectra_rccl(engine, hdr, blk, n_genes):
    """The gather through the C-ABI library: one small ``cnmf_allgather_bytes`` for the sizes,
    one for the headers, then ONE ``cnmf_allgather_spectra`` (ncclAllGather) for the payload.
    ``blk=None`` sends the context's resident spectra store (rows in ``hdr`` order)."""
    world = engine.comm_world
    rows_local = engine.spectra_rows if blk is None else int(blk.shape[0])
    sizes = engine.allgather_array(np.array([hdr.shape[0], rows_local], dtype=np.int64))
    max_n = max(1, int(sizes[:, 0].max()))
    max_rows = max(1, int(sizes[:, 1].max()))
    h = np.zeros((max_n, 3), dtype=np.int32)
    h[:hdr.shape[0]] = hdr
    hs = engine.allgather_array(h)
    bs = engine.allgather_spectra(blk, max_rows, n_genes)
    return unpack([hs[r, :int(sizes[r, 0])] for r in range(world)],
                  [bs[r, :int(sizes[r, 1])] for r in range(world)])


def unpack(headers, blocks):
    out = {}
    for hdr, blk in zip(headers, blocks):
        off = 0
        for _, k, it in hdr:
            out[(int(k), int(it))] = blk[off:off + int(k)]
            off += int(k)
    return out


def factorize_distributed(obj, rank, world, device=None, gather="rccl", **factorize_kwargs):
    """``cNMF.factorize`` on this rank's shard, then the gather: afterwards every rank's
    ``obj.spectra_cache`` holds every restart that was run, so ``obj.combine()`` needs no files.
    ``gather="rccl"`` (default) uses the library's own communicator (``obj.engine`` must have had
    ``comm_init``, e.g. via ``comm_bootstrap_file``); ``"torch"`` uses torch.distributed (gloo in the CPU
    tests).  The rows exchanged are the ones ``factorize`` actually ran (``obj.last_factorize_jobs``): with
    ``skip_completed_runs=True`` the shard is taken over the INCOMPLETE rows (cnmf.py:729-733), not over the
    whole ledger."""
    import pandas as pd
    from .cnmf import load_df_from_npz
    run_params = load_df_from_npz(obj.paths["nmf_replicate_parameters"])
    obj.factorize(worker_i=rank, total_workers=world, **factorize_kwargs)
    rows, spectra = [], []
    genes = None
    for idx in obj.last_factorize_jobs:
        p = run_params.loc[idx]
        key = (int(p["n_components"]), int(p["iter"]))
        if key in obj.spectra_cache:
            genes = obj._spectra_columns
            rows.append((idx, key[0], key[1]))
            spectra.append(np.asarray(obj.spectra_cache[key]))
    if genes is None:
        genes = load_df_from_npz(obj.paths["normalized_counts"]).columns
    hdr, blk = pack_local(rows, spectra, len(genes))
    if gather == "rccl":
        merged = allgather_spectra_rccl(obj.engine, hdr, blk, len(genes))
    else:
        merged = allgather_spectra(hdr, blk, len(genes), device=device)
    for (k, it), H in merged.items():
        obj.spectra_cache[(k, it)] = H.astype(np.float64)
    obj._spectra_columns = genes
    return merged
