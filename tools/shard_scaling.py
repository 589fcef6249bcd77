"""Projected strong-scaling curve of the north-star job from ONE GPU: run, for world = 1, 2, 4, 8, only the shard that
rank 0 of that world would own (`bench.py --emulate-rank 0/W`: ledger rows idx % W == 0 of the 900-restart job, the
reference's worker_filter, cnmf.py:52-53) and report restarts/s of the shard, its column utilisation and the time
spent after the queue of pending restarts ran dry (the tail).  With no collective inside the restart loop and one
latency-bound all-gather (<= 16 MB per GPU) at the end, the job's time on W GPUs is the slowest shard's time:
    projected efficiency(W) = (restarts of the job / W) / shard_seconds(W) / restarts_per_s(1).
Writes gpurun_out/shard_scaling.json.   python tools/shard_scaling.py [--steps 2] [--restarts-per-k 100]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--restarts-per-k", type=int, default=100)
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--ranks", default="0", help="ranks of each world to emulate (comma separated; 'all' = every rank)")
    a = ap.parse_args()
    rows = []
    for W in [int(w) for w in a.worlds.split(",")]:
        ranks = range(W) if a.ranks == "all" else [int(r) for r in a.ranks.split(",") if int(r) < W]
        for r in ranks:
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", str(a.steps), "--warmup", str(a.warmup),
                   "--restarts-per-k", str(a.restarts_per_k), "--no-cpu-baseline", "--no-extras", "--emulate-rank", "%d/%d" % (r, W)]
            p = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT)
            if p.returncode != 0:
                raise SystemExit("bench failed for %d/%d: %s" % (r, W, p.stderr[-2000:]))
            d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
            c = d["config"]
            rows.append({"world": W, "rank": r, "restarts_per_step": c["restarts_per_step_per_gpu"],
                         "shard_seconds_per_step": d["ms_per_step"] / 1e3, "shard_restarts_per_s": d["value"],
                         "column_utilisation": c["column_utilisation"], "mean_iterations_per_restart": c["mean_iterations_per_restart"],
                         "tail_ms_per_step": c["tail"]["ms_per_step"], "tail_share_of_gpu_time": c["tail"]["share_of_gpu_time"],
                         "tail_mean_live_columns": c["tail"]["mean_live_columns"],
                         "gemm_share_of_gpu_time": d["roofline"]["gemm_share_of_gpu_time"]})
            print(json.dumps(rows[-1]), flush=True)
    base = [x for x in rows if x["world"] == 1]
    job = 9 * a.restarts_per_k
    proj = []
    if base:
        r1 = base[0]["shard_restarts_per_s"]
        for W in sorted({x["world"] for x in rows}):
            slow = max(x["shard_seconds_per_step"] for x in rows if x["world"] == W)
            proj.append({"world": W, "job_seconds": slow, "job_restarts_per_s": job / slow, "projected_efficiency": job / slow / (W * r1)})
    from bench import source_hashes
    out = {"_source": "tools/shard_scaling.py: bench.py --emulate-rank R/W on ONE MI355X (strong scaling of the %d-restart "
                      "north-star job projected from single-GPU shards; the all-gather at the end -- <= 16 MB per GPU, "
                      "latency-bound on xGMI -- is not included)" % job,
           "shards": rows, "projection": proj, "kernel_source_sha256": source_hashes()}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "shard_scaling.json"), "w"), indent=1)
    print(json.dumps(proj, indent=1))


if __name__ == "__main__":
    main()
