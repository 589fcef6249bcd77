"""Offline simulation of the restart scheduler (batch_host.hip.h::run_batch) on MEASURED iteration counts
(gpurun_out/iters_c3.json from tools/dump_iters.py): queue orders and batch widths for the north-star job and its
strong-scaling shards.  Cost model per batch iteration (microseconds, measured at C3 on one MI355X, round 3):
T(256) = 270, T(512) = 450, T(1024) = 800 when the batch is full; the tail narrows in 256-column steps."""
import json, sys
import numpy as np

T_FULL = {256: 270.0, 512: 450.0, 768: 640.0, 1024: 800.0}


def simulate(ks, its, KC, order="lpt_mean", narrow=True, tail_cost=None):
    n = len(ks)
    if order == "rank":
        q = sorted(range(n), key=lambda r: -ks[r])
    elif order == "oracle":
        q = sorted(range(n), key=lambda r: -its[r])
    else:
        by = {}
        for r in range(n):
            by.setdefault(ks[r], []).append(r)
        q, j = [], 0
        while len(q) < n:
            for k in sorted(by, reverse=True):
                if j < len(by[k]):
                    q.append(by[k][j])
            j += 1
    live = {}            # restart -> remaining iterations
    free = KC
    t = 0.0
    done_stats = {}      # k -> [sum, count, max]
    n_done_at_resort = 0
    n_done = 0
    tail_t = 0.0
    iters = 0
    while q or live:
        # refill (first fit in queue order)
        i = 0
        while i < len(q):
            r = q[i]
            if ks[r] <= free:
                live[r] = its[r]; free -= ks[r]; q.pop(i)
            else:
                i += 1
                if free < 5:
                    break
        cols = sum(ks[r] for r in live)
        kc_now = KC
        if not q and narrow:
            kc_now = max(256, -(-cols // 256) * 256)
        step = min(live.values())          # advance until the next retirement
        cost = T_FULL[kc_now] if tail_cost is None or q else tail_cost(kc_now, cols)
        t += step * cost
        iters += step
        if not q:
            tail_t += step * cost
        for r in list(live):
            live[r] -= step
            if live[r] == 0:
                del live[r]; free += ks[r]; n_done += 1
                s = done_stats.setdefault(ks[r], [0, 0, 0]); s[0] += its[r]; s[1] += 1; s[2] = max(s[2], its[r])
        if order in ("lpt_mean", "lpt_max", "lpt_p") and q and n_done - n_done_at_resort >= 8:
            n_done_at_resort = n_done
            def exp(r):
                s = done_stats.get(ks[r])
                if not s:
                    return 1e30
                return s[0] / s[1] if order == "lpt_mean" else (s[2] + 1e-3 * s[0] / s[1])
            q.sort(key=lambda r: (-exp(r), -ks[r]))
    return n / (t * 1e-6), tail_t * 1e-3, t * 1e-3


def main():
    d = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/iters_c3.json"))
    ks_all, it_all, iters = np.array(d["k"]), np.array(d["iter"]), np.array(d["n_iter"])
    sel = it_all < 100                                   # the north-star job: n_iter = 100
    ks, its = list(ks_all[sel]), list(iters[sel])
    # partial-tile skipping in the tail: cost ~ max(HBM floor, live share)
    def tail_skip(kc, cols):
        full = T_FULL[kc]
        return max(0.45 * 270.0 * kc / 256 if cols > 128 else 140.0, full * max(cols, 1) / kc * 0.85 + 60.0)
    for W in (1, 2, 4, 8):
        idx = [i for i in range(len(ks)) if i % W == 0]
        k_s, i_s = [ks[i] for i in idx], [its[i] for i in idx]
        ideal = sum(a * b for a, b in zip(k_s, i_s))
        print("world %d: %d restarts, %d column-iterations" % (W, len(idx), ideal))
        for KC in (256, 512, 1024):
            row = []
            for order in ("rank", "lpt_mean", "lpt_max", "oracle"):
                r, tail, tot = simulate(k_s, i_s, KC, order)
                row.append("%s %.1f/s (tail %.0f of %.0f ms)" % (order, r, tail, tot))
            r, tail, tot = simulate(k_s, i_s, KC, "lpt_mean", tail_cost=tail_skip)
            row.append("lpt_mean+skip %.1f/s (tail %.0f)" % (r, tail))
            print("   KC %4d: " % KC + " | ".join(row))


if __name__ == "__main__":
    main()
