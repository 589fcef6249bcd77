"""A/B of the W half-step inside pass A (kernels_fusedw.hip.h) against the stand-alone sweep: the same batch under
CNMF_FUSE_A=1 (opt-in: measured slower) and CNMF_FUSE_A=2 (the stand-alone sweep with the same
per-tile partials -- the default adds its partials per workgroup of four tiles, a different rounding) in two processes (the switch is read once per process); spectra, usages, iteration
counts and violations must be BIT-IDENTICAL.  `python tools/fused_ab.py child <tag>` runs one arm."""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(tag):
    from cnmf_amd import synth
    from cnmf_amd.engine import Engine
    n_cells = int(os.environ.get("AB_CELLS", "50000"))
    X = synth.make_config("C3", dtype=np.float32, n_cells=n_cells)
    rs = np.random.RandomState(5)
    n = int(os.environ.get("AB_RESTARTS", "130"))
    ks = [int(k) for k in rs.randint(5, 14, size=n)]
    if os.environ.get("AB_MIXED"):
        ks[3], ks[17], ks[40] = 20, 33, 16                       # ranks the epilogue does not take (tiers 1, 2) and the largest it does
    seeds = [int(s) for s in rs.randint(1, 2 ** 31 - 1, size=n)]
    with Engine(0) as eng:
        eng.set_matrix(X)
        eng.nmf_batch(ks[:4], seeds=seeds[:4], max_iter=3, warn=False)      # warm-up
        t0 = time.perf_counter()
        H, W, n_iter, viol = eng.nmf_batch(ks, seeds=seeds, max_iter=int(os.environ.get("AB_ITERS", "40")), warn=False,
                                           return_W=bool(os.environ.get("AB_W")))
        dt = time.perf_counter() - t0
        st = eng.last_stats
    out = os.path.join(ROOT, "gpurun_out", "fused_ab_%s.npz" % tag)
    np.savez(out, H=np.concatenate([h.ravel() for h in H]), n_iter=n_iter, viol=viol,
             W=(np.concatenate([w.ravel() for w in W]) if W is not None else np.zeros(1)))
    print(json.dumps({"tag": tag, "seconds": dt, "kc": int(st["kc"]), "gemm_mode": int(st["gemm_mode"]),
                      "outer_iterations": int(st["outer_iterations"]), "gpu_ms": st["gpu_ms"]}))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        return child(sys.argv[2])
    res = {}
    for tag, val in (("fused", "1"), ("plain", "2")):       # 2: the stand-alone sweep with the same per-tile partials
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "child", tag],
                           env=dict(os.environ, CNMF_FUSE_A=val, CNMF_DEBUG="1"), capture_output=True, text=True, timeout=900)
        if p.returncode != 0:
            print(p.stdout[-2000:], p.stderr[-4000:])
            raise SystemExit("arm %s failed" % tag)
        res[tag] = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
        fused_lines = [ln for ln in p.stderr.splitlines() if "W half-step inside pass A" in ln]
        res[tag]["fused_iterations"] = int(fused_lines[-1].split(":")[1].split("of")[0]) if fused_lines else -1
        print(res[tag])
    a = np.load(os.path.join(ROOT, "gpurun_out", "fused_ab_fused.npz"))
    b = np.load(os.path.join(ROOT, "gpurun_out", "fused_ab_plain.npz"))
    ok = True
    for key in ("H", "W", "n_iter", "viol"):
        same = np.array_equal(a[key], b[key])
        md = float(np.abs(a[key].astype(np.float64) - b[key].astype(np.float64)).max())
        print("%-7s identical: %s  (max abs diff %.3g, max |ref| %.3g, %d of %d differ)"
              % (key, same, md, float(np.abs(b[key]).max()), int((a[key] != b[key]).sum()), a[key].size))
        ok &= same
    ok &= res["fused"]["fused_iterations"] > 0 and res["plain"]["fused_iterations"] == 0        # the arms really differ
    print("FUSED_AB_%s  fused %.3f s (%d iterations with the epilogue)  plain %.3f s" % ("IDENTICAL" if ok else "DIFFERENT",
          res["fused"]["seconds"], res["fused"]["fused_iterations"], res["plain"]["seconds"]))
    for tag in ("fused", "plain"):
        os.remove(os.path.join(ROOT, "gpurun_out", "fused_ab_%s.npz" % tag))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
