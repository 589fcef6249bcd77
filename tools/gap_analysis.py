"""Inter-kernel gaps of the restart loop from a rocprofv3 rocpd trace (gpurun_out/prof*/trace_results.db)."""
import collections, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, end from kernels order by start"))
key = sys.argv[2] if len(sys.argv) > 2 else "gemm"
idx = [i for i, r in enumerate(rows) if key in r[0]]
i0, i1 = idx[len(idx) // 10], idx[-len(idx) // 5]
seg = rows[i0:i1]
busy = sum(r[2] - r[1] for r in seg); wall = seg[-1][2] - seg[0][1]
print("kernels %d busy %.1f ms wall %.1f ms busy frac %.4f" % (len(seg), busy / 1e6, wall / 1e6, busy / wall))
gaps = collections.defaultdict(list)
for a, b in zip(seg[:-1], seg[1:]):
    gaps[(a[0].split('(')[0][-30:], b[0].split('(')[0][-30:])].append(b[1] - a[2])
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:8]:
    print("  %-32s -> %-32s n=%5d mean gap %.2f us total %.1f ms" % (k[0], k[1], len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6))
