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


def test_rccl_init_failure_exits_non_zero_on_every_rank_by_default():
    """Round-5 review, item 6: the first real N > 1 run must not be able to turn "the in-library RCCL communicator did not
    form" into a green number.  The set-up fails on rank 1 of 3: the ranks agree on that (status files beside the RCCL id),
    EVERY rank exits non-zero, stdout carries no JSON line."""
    p = _run(["--gpus", "3", "--spawn-selftest"], env={"CNMF_BENCH_SELFTEST_RCCL_FAIL_RANKS": "1"})
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert "did not form on 1 of 3 ranks" in p.stderr and "simulated failure on rank 1" in p.stderr
    assert "No number is reported" in p.stderr


def test_rccl_init_failure_with_explicit_fallback_reports_under_another_metric():
    """--allow-transport-fallback: ALL ranks switch together (no rank keeps a communicator the others left), and the
    line's metric cannot be mistaken for the headline."""
    p = _run(["--gpus", "2", "--spawn-selftest", "--allow-transport-fallback"], env={"CNMF_BENCH_SELFTEST_RCCL_FAIL_RANKS": "0"})
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert d["gather"] == "torch" and "rank 0" in d["gather_fallback"]
    assert "TRANSPORT FALLBACK" in d["metric"] and not d["metric"].startswith("NMF restarts/sec (")


def test_every_rank_takes_the_same_transport_decision(tmp_path):
    """The healthy ranks must not go on alone over a communicator the failed rank never joined (round-5 advice): with the
    fallback allowed ALL three return "torch" with the same reason; without it ALL three stop."""
    import importlib.util
    import threading
    spec = importlib.util.spec_from_file_location("bench_mod", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    def run(allow, sub):
        res = [None] * 3

        def one(rank):
            def boot():
                if rank == 2:
                    raise RuntimeError("boom")
            try:
                res[rank] = bench.form_transport(boot, rank, 3, str(tmp_path / sub / "id"), allow, timeout=20.0)
            except SystemExit as e:
                res[rank] = ("exit", str(e))
        os.makedirs(tmp_path / sub)
        th = [threading.Thread(target=one, args=(r,)) for r in range(3)]
        [t.start() for t in th]
        [t.join() for t in th]
        return res
    res = run(True, "a")
    assert [r[0] for r in res] == ["torch"] * 3 and len({r[1] for r in res}) == 1 and "rank 2" in res[0][1]
    res = run(False, "b")
    assert [r[0] for r in res] == ["exit"] * 3 and all("did not form on 1 of 3 ranks" in r[1] for r in res)


def test_rccl_init_success_keeps_the_library_transport_and_the_headline_metric():
    p = _run(["--gpus", "2", "--spawn-selftest"], env={"CNMF_BENCH_SELFTEST_RCCL_FAIL_RANKS": ""})
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads(p.stdout)
    assert d["gather"] == "rccl" and d["gather_fallback"] is None and d["metric"].startswith("NMF restarts/sec (")


def test_agree_on_outcome_times_out_naming_the_silent_ranks(tmp_path):
    """A rank stuck inside ncclCommInitRank never reports: the others stop with its name instead of waiting for the
    launcher's timeout."""
    import pytest
    sys.path.insert(0, ROOT)
    from cnmf_amd import dist as cd
    with pytest.raises(TimeoutError, match=r"ranks \[1, 2\] never reported"):
        cd.agree_on_outcome(str(tmp_path / "rccl_id"), 0, 3, True, timeout=0.3)
