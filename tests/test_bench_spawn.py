"""`python bench.py --gpus N` must start its N ranks ITSELF (no torch.distributed.run, no torch) and must never report
a number under the wrong n_gpus.  The launch plumbing is exercised here without a GPU through the hidden
``--spawn-selftest`` mode (every rank only reports the environment it was given); the GPU side -- the in-library RCCL
rendezvous of the spawned ranks -- is covered by tests/test_gpu_comm.py."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=120):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "CNMF_RCCL_ID_FILE"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=e)


def test_gpus_n_spawns_n_ranks_and_relays_one_json_line():
    p = _run(["--gpus", "4", "--spawn-selftest"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1                                   # exactly ONE JSON line on stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["rank"] == 0 and d["local_rank"] == 0 and d["spawned"] == "1"
    # the RCCL id travels through a file in the launch's private directory, which is gone afterwards
    assert os.path.basename(d["id_file"]) == "rccl_id" and "cnmf_launch_" in d["id_file"]
    assert not os.path.exists(os.path.dirname(d["id_file"]))
    assert d["master"].startswith("127.0.0.1:")
    assert "torch" not in p.stderr.lower()


def test_a_failing_rank_fails_the_run_without_a_result():
    p = _run(["--gpus", "3", "--spawn-selftest"], env={"CNMF_BENCH_SELFTEST_FAIL_RANK": "2"})
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert "rank 2 of 3 exited with code 3" in p.stderr


def test_world_size_mismatch_is_refused():
    """An external launcher that started 2 ranks for `--gpus 8` (or a stray WORLD_SIZE) must not produce a line
    labelled with either number."""
    p = _run(["--gpus", "8", "--spawn-selftest"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "refusing" in p.stderr and not p.stdout.strip()
    # under a launcher whose world matches, the rank runs as that launcher's rank (no second spawn)
    p = _run(["--gpus", "2", "--spawn-selftest"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode == 0 and json.loads(p.stdout)["spawned"] is None


def test_bench_has_no_torch_in_its_default_path():
    src = open(BENCH).read()
    assert "import torch" not in src.split("def main")[0]                  # nothing at module level, nothing in the spawner
    code = [ln for ln in src.splitlines() if "torch.distributed.run" in ln and not ln.lstrip().startswith(("#", "launcher", "\"", "rank"))]
    assert all("Popen" not in ln and "subprocess" not in ln for ln in code)  # mentioned in prose only, never executed
