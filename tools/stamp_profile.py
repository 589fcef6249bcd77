"""Stamp a measured profile (JSON written on the GPU box under gpurun_out/) while copying it into profiles/: adds the
git HEAD it was measured at and -- unless the measuring script already did -- the sha256 of the kernel sources it
describes, so that bench.py can refuse a profile that no longer matches the tree (bench.load_profile).
    python tools/stamp_profile.py gpurun_out/pmc_traffic.json profiles/r3_pmc_traffic_f16.json"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from bench import source_hashes
    src, dst = sys.argv[1], sys.argv[2]
    d = json.load(open(src))
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
    dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "cnmf_amd/csrc"], capture_output=True, text=True).stdout.strip()
    d["git_head"] = head + ("+uncommitted csrc changes" if dirty else "")
    now = source_hashes()
    if "kernel_source_sha256" in d and d["kernel_source_sha256"] != now:
        print("WARNING: the profile was measured on other kernel sources than the tree holds now:", file=sys.stderr)
        for k, v in d["kernel_source_sha256"].items():
            if now.get(k) != v:
                print("   ", k, v, "->", now.get(k), file=sys.stderr)
    d.setdefault("kernel_source_sha256", now)
    json.dump(d, open(dst, "w"), indent=1)
    print("stamped", dst, d["git_head"])


if __name__ == "__main__":
    main()
