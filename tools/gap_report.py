"""Inter-kernel gaps of the main stream in a rocprofv3 kernel trace (rocpd database): per (kernel -> next kernel) pair,
median / mean gap in the steady-state middle third of the run."""
import sqlite3, sys, collections
import numpy as np
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name,start,end from kernels order by start"))
rows = rows[len(rows) // 3: 2 * len(rows) // 3]
gaps = collections.defaultdict(list)
for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
    a = n0.split('(')[0].replace('void cnmf::', '')[:30]; b = n1.split('(')[0].replace('void cnmf::', '')[:30]
    gaps[(a, b)].append((s1 - e0) / 1e3)
for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:8]:
    v = np.array(v)
    print('%-32s -> %-32s n=%5d median %6.2f us  mean(<100us) %6.2f' % (k[0], k[1], len(v), np.median(v), v[v < 100].mean()))
