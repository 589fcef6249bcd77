"""Export the per-kernel summary (the `--stats` view) of a rocprofv3 rocpd sqlite trace
into a small text file under profiles/ (the .db itself is scratch)."""
import sqlite3
import sys


def main(db, out, title=""):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    with open(out, "w") as f:
        f.write("# %s\n# source: rocprofv3 --kernel-trace --stats (rocpd sqlite `top_kernels` view); durations in microseconds\n" % title)
        f.write("%-110s %8s %14s %12s %8s\n" % ("kernel", "calls", "total_us", "avg_us", "pct"))
        for n, calls, tot, avg, pct in rows:
            f.write("%-110s %8d %14.0f %12.1f %7.2f%%\n" % (n[:110], calls, tot, avg, pct))
    print(open(out).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")
