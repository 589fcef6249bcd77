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

Round 4 -- the N-rank launcher lives here, in the product (``launch_ranks`` / ``factorize_multi_gpu`` /
``python -m cnmf_amd.dist worker ...``): the counterpart of the reference's ``factorize_multi_process``
(cnmf.py:677-689, ``factorize_mp_signature`` :254-262: a multiprocessing Pool of workers, each running
``factorize(worker_i, total_workers)``).  One process per GPU, file rendezvous, one all-gather, rank 0 combines; the
launcher polls every rank, names the stage a dead rank reached, and kills exactly the processes it started.
"""
import json
import os
import subprocess
import sys
import time

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


def agree_on_outcome(path, rank, world, ok, message="", timeout=120.0, poll=0.05):
    """Every rank reports whether a collective set-up step succeeded for IT, then waits for the reports of all ranks, so that
    all of them take the SAME decision afterwards (a communicator that formed on a subset of the ranks must not be used
    by that subset while the others go elsewhere: that is a dead-lock until the launcher's timeout, not an error message).
    Reports travel like the RCCL id does -- files ``<path>.status.<rank>`` written atomically.  Returns
    ``[(rank, ok, message), ...]`` for all ranks; raises ``TimeoutError`` naming the ranks that never reported (stuck
    inside the collective, or dead)."""
    tmp = "%s.status.%d.tmp.%d" % (path, rank, os.getpid())
    with open(tmp, "w") as f:
        json.dump({"ok": bool(ok), "message": str(message)}, f)
    os.replace(tmp, "%s.status.%d" % (path, rank))
    t0 = time.time()
    reports = {}
    while len(reports) < world:
        for r in range(world):
            if r in reports:
                continue
            try:
                with open("%s.status.%d" % (path, r)) as f:
                    reports[r] = json.load(f)
            except (OSError, ValueError):
                pass
        if len(reports) < world:
            if time.time() - t0 > timeout:
                missing = [r for r in range(world) if r not in reports]
                raise TimeoutError("ranks %s never reported the outcome of the communicator set-up within %.0f s "
                                   "(stuck inside the collective, or dead)" % (missing, timeout))
            time.sleep(poll)
    return [(r, bool(reports[r]["ok"]), reports[r]["message"]) for r in range(world)]


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
        genes = obj._load_norm_counts().columns
    hdr, blk = pack_local(rows, spectra, len(genes))
    if gather == "rccl":
        merged = allgather_spectra_rccl(obj.engine, hdr, blk, len(genes))
    else:
        merged = allgather_spectra(hdr, blk, len(genes), device=device)
    for (k, it), H in merged.items():
        obj.spectra_cache[(k, it)] = H.astype(np.float64)
    obj._spectra_columns = genes
    return merged


# ------------------------------------------------------------------------------------------ the N-rank launcher
class RankFailure(RuntimeError):
    """A rank of a multi-process launch died (or the launch timed out); the message names rank, exit code, the stage
    every rank had reached and the tail of the failing rank's stderr."""


_STAGES = ("spawned", "imported", "engine", "rendezvous", "comm_ready", "factorize", "gathered", "combined", "done")


def _stage_path(run_dir, rank):
    return os.path.join(run_dir, "rank%d.stage" % rank)


def mark_stage(stage):
    """Called by a rank: record how far it got (one small file per rank under CNMF_LAUNCH_DIR; a no-op outside a
    launch) -- what the launcher quotes when a rank dies, e.g. "rank 3 died at 'engine'; ranks 0-2 were waiting in
    'rendezvous'": a rank that never reaches ``ncclCommInitRank`` would otherwise leave the others inside it for ever."""
    d = os.environ.get("CNMF_LAUNCH_DIR")
    if not d:
        return
    try:
        with open(_stage_path(d, int(os.environ.get("RANK", "0"))), "w") as f:
            f.write(stage)
    except OSError:
        pass


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(argv, n, env=None, timeout=None, poll=0.1, capture_stdout=True):
    """Start ``n`` processes running ``argv`` -- rank r with RANK / LOCAL_RANK = r, WORLD_SIZE = n, a private launch
    directory (CNMF_LAUNCH_DIR: stage files, stderr logs, the RCCL id file CNMF_RCCL_ID_FILE) -- and wait for all of them.
    Returns rank 0's stdout (bytes).  If any rank exits non-zero, or ``timeout`` seconds pass, every process started
    here (exactly those PIDs) is killed and ``RankFailure`` says which rank failed, where each rank was, and why."""
    import tempfile
    run_dir = tempfile.mkdtemp(prefix="cnmf_launch_")
    port = _free_port()
    base = dict(os.environ if env is None else env)
    procs, logs = [], []
    out0 = b""
    failure = None
    t0 = time.time()
    try:
        for r in range(n):
            e = dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                     MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CNMF_LAUNCH_DIR=run_dir,
                     CNMF_RCCL_ID_FILE=os.path.join(run_dir, "rccl_id"),
                     HSA_ENABLE_IPC_MODE_LEGACY=base.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            log = open(os.path.join(run_dir, "rank%d.stderr" % r), "wb")
            logs.append(log)
            with open(_stage_path(run_dir, r), "w") as f:
                f.write("spawned")
            procs.append(subprocess.Popen(list(argv), env=e, stdout=subprocess.PIPE if (r == 0 and capture_stdout) else log,
                                          stderr=log))
        import selectors
        sel = selectors.DefaultSelector()
        open_out = capture_stdout
        if open_out:
            sel.register(procs[0].stdout, selectors.EVENT_READ)
        while True:
            if open_out:
                for key, _ in sel.select(timeout=poll):
                    chunk = os.read(key.fileobj.fileno(), 65536)
                    if chunk:
                        out0 += chunk
                    else:
                        sel.unregister(key.fileobj)
                        open_out = False
            else:
                time.sleep(poll)
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                failure = ("exit", bad[0][0], bad[0][1])
                break
            if all(c == 0 for c in codes):
                if open_out:
                    # every rank has exited: drain what rank 0 wrote and stop -- a grandchild that inherited the pipe
                    # must not keep the parent waiting
                    try:
                        os.set_blocking(procs[0].stdout.fileno(), False)
                        while True:
                            chunk = os.read(procs[0].stdout.fileno(), 65536)
                            if not chunk:
                                break
                            out0 += chunk
                    except (BlockingIOError, OSError):
                        pass
                break
            if timeout is not None and time.time() - t0 > timeout:
                failure = ("timeout", None, None)
                break
    finally:
        if failure is not None:
            time.sleep(0.5)                               # let the failing rank's message reach its log
        for p in procs:
            if p.poll() is None:
                p.kill()                                  # exactly the PIDs started here
        for p in procs:
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                p.kill()
        for log in logs:
            log.close()
    stages = []
    for r in range(n):
        try:
            stages.append(open(_stage_path(run_dir, r)).read().strip())
        except OSError:
            stages.append("?")
    if failure is not None:
        kind, r, code = failure
        def tail(rr):
            try:
                return open(os.path.join(run_dir, "rank%d.stderr" % rr), "rb").read()[-2000:].decode(errors="replace")
            except OSError:
                return ""
        where = ", ".join("rank %d: %r" % (i, st) for i, st in enumerate(stages))
        if kind == "exit":
            msg = ("rank %d of %d exited with code %d at stage %r -- the launch was stopped and the other ranks killed "
                   "(they would have waited for it inside the collective for ever).  Stages reached: %s.\n--- stderr of "
                   "rank %d ---\n%s" % (r, n, code, stages[r], where, r, tail(r)))
        else:
            slow = min(range(n), key=lambda i: _STAGES.index(stages[i]) if stages[i] in _STAGES else -1)
            msg = ("no result after %.0f s -- the launch was stopped and all ranks killed.  Stages reached: %s.\n--- stderr "
                   "of rank %d ---\n%s" % (timeout, where, slow, tail(slow)))
        _rmtree_quiet(run_dir)
        raise RankFailure(msg)
    _rmtree_quiet(run_dir)
    return out0


def _rmtree_quiet(d):
    import shutil
    shutil.rmtree(d, ignore_errors=True)


def _load_factory(spec):
    """"module:callable" -> the callable (the engine factory of a worker: ``factory(local_rank) -> engine``)."""
    import importlib
    mod, _, fn = spec.partition(":")
    return getattr(importlib.import_module(mod), fn)


DEFAULT_LAUNCH_TIMEOUT = 24 * 3600.0      # a hung rank (ncclCommInitRank, the all-gather) must not block the parent for ever


def factorize_multi_gpu(obj, n_gpus=None, skip_completed_runs=False, gather="rccl", write_iter_files=True,
                        timeout=DEFAULT_LAUNCH_TIMEOUT, engine_factory=None, python=None):
    """The mirror of the reference's ``cNMF.factorize_multi_process(total_workers)`` (cnmf.py:677-689) for GPUs: spawn
    ``n_gpus`` processes (default: every visible GPU), rank r on GPU r running ``factorize(worker_i=r,
    total_workers=n_gpus)`` -- the reference's own shard, ``worker_filter`` (cnmf.py:52-53) -- then

    * ``gather="rccl"`` (default): ONE all-gather of the spectra through the library's communicator (file rendezvous of
      the 128-byte id, ``ncclCommInitRank``, ``cnmf_allgather_spectra``); rank 0 writes the merged-spectra files
      (``combine``), so the parent finds what ``combine()`` would have produced;
    * ``gather="files"``: the reference's own gather -- every rank writes its per-iteration files, the parent combines
      from them (``combine()``); no collective at all.

    ``write_iter_files``: keep the reference's per-iteration files too (always on for ``gather="files"``).
    ``engine_factory`` ("module:callable", test hook): what a worker calls instead of ``Engine(local_rank)``.
    Raises :class:`RankFailure` when a rank dies or ``timeout`` seconds pass (default: a day -- a rank that hangs without
    dying is otherwise never detected; ``None`` waits for ever).  Multi-GPU execution of the RCCL path
    has not been measured on hardware yet."""
    if n_gpus is None:
        from . import _lib
        n_gpus = max(1, int(_lib.load().cnmf_device_count()))
    if gather not in ("rccl", "files"):
        raise ValueError("gather must be 'rccl' or 'files'")
    argv = [python or sys.executable, "-m", "cnmf_amd.dist", "worker", "--output-dir", obj.output_dir, "--name", obj.name,
            "--gather", gather]
    if skip_completed_runs:
        argv.append("--skip-completed-runs")
    if not write_iter_files and gather != "files":
        argv.append("--no-iter-files")
    if engine_factory:
        argv += ["--engine-factory", engine_factory]
    if not getattr(obj, "detect_counts", True):
        argv.append("--no-count-detection")
    env = dict(os.environ)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = root + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
    out = launch_ranks(argv, int(n_gpus), env=env, timeout=timeout)
    report = [json.loads(ln) for ln in out.decode(errors="replace").splitlines() if ln.startswith("{")]
    # what the workers wrote supersedes whatever an earlier factorize() of THIS process left in memory (another shard,
    # another batch placement): combine() must read the files, consensus must not map rows into this process's old store
    if hasattr(obj, "spectra_cache"):
        obj.spectra_cache.clear()
    if hasattr(obj, "_store_rows"):
        obj._store_rows.clear()
    if gather == "files":
        obj.combine()
    obj.merged_cache.clear()                              # the merged files were (re)written by another process
    return report[-1] if report else {}


def _worker_main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(prog="python -m cnmf_amd.dist")
    ap.add_argument("mode", choices=("worker",))
    ap.add_argument("--output-dir", required=True)
    ap.add_argument("--name", required=True)
    ap.add_argument("--gather", choices=("rccl", "files"), default="rccl")
    ap.add_argument("--skip-completed-runs", action="store_true")
    ap.add_argument("--no-iter-files", action="store_true")
    ap.add_argument("--no-count-detection", action="store_true")
    ap.add_argument("--engine-factory", default=None)
    ap.add_argument("--rendezvous-timeout", type=float, default=300.0)
    a = ap.parse_args(argv)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    mark_stage("imported")
    from .cnmf import cNMF
    engine = _load_factory(a.engine_factory)(local) if a.engine_factory else None
    obj = cNMF(output_dir=a.output_dir, name=a.name, device=local, engine=engine, detect_counts=not a.no_count_detection)
    eng = obj.engine                                       # creates the device context: fails loudly without a GPU
    mark_stage("engine")
    t0 = time.time()
    if a.gather == "rccl":
        mark_stage("rendezvous")
        id_file = os.environ.get("CNMF_RCCL_ID_FILE") or os.path.join(a.output_dir, a.name, "cnmf_tmp", "rccl_id")
        comm_bootstrap_file(eng, rank, world, id_file, timeout=a.rendezvous_timeout)
        mark_stage("comm_ready")
        mark_stage("factorize")
        merged = factorize_distributed(obj, rank, world, gather="rccl", skip_completed_runs=a.skip_completed_runs,
                                       write_iter_files=not a.no_iter_files)
        mark_stage("gathered")
        n_total = len(merged)
        if rank == 0:
            obj.combine(skip_missing_files=a.skip_completed_runs)
            mark_stage("combined")
    else:
        mark_stage("factorize")
        obj.factorize(worker_i=rank, total_workers=world, skip_completed_runs=a.skip_completed_runs, write_iter_files=True)
        n_total = len(obj.last_factorize_jobs)
    if rank == 0:
        sys.stdout.write(json.dumps({"world": world, "gather": a.gather, "restarts_seen_by_rank0": int(n_total),
                                     "rank0_seconds": time.time() - t0}) + "\n")
        sys.stdout.flush()
    mark_stage("done")


if __name__ == "__main__":
    _worker_main()
