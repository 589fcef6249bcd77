"""Kernel-overlap report of a rocprofv3 --kernel-trace rocpd database: how much of the GPU time had two kernels
(of different streams) in flight, per-kernel average durations, and a sample of the timeline."""
import sqlite3, sys, collections
db = sys.argv[1]
c = sqlite3.connect(db)
cur = c.execute("select * from kernels limit 1")
cols = [d[0] for d in cur.description]
print("columns:", cols)
qcol = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
rows = list(c.execute("select name, start, end, %s from kernels order by start" % (qcol or "0")))
t0 = rows[0][1]
by = collections.defaultdict(list)
for n, s, e, q in rows:
    by[n.split("(")[0][-60:]].append((e - s) / 1e3)
print("%d dispatches, streams/queues: %s" % (len(rows), sorted(set(r[3] for r in rows))))
# sweep line: time with >= 1 and >= 2 kernels in flight
ev = []
for n, s, e, q in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy1 = busy2 = 0; depth = 0; last = ev[0][0]
for t, d in ev:
    if depth >= 1: busy1 += t - last
    if depth >= 2: busy2 += t - last
    depth += d; last = t
tot = sum(e - s for _, s, e, _ in rows)
print("span %.1f ms, busy(>=1) %.1f ms, busy(>=2) %.1f ms, sum of durations %.1f ms" % ((rows[-1][2] - t0) / 1e6, busy1 / 1e6, busy2 / 1e6, tot / 1e6))
for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print("%-62s n=%6d avg %8.1f us" % (n, len(v), sum(v) / len(v)))
mid = len(rows) * 2 // 3
print("timeline sample (us from first shown):")
b = rows[mid][1]
for n, s, e, q in rows[mid:mid + int(sys.argv[2]) if len(sys.argv) > 2 else mid + 40]:
    print("  q%-3s %9.1f -> %9.1f  %s" % (q, (s - b) / 1e3, (e - b) / 1e3, n.split("(")[0][-50:]))
