"""Dump the kernel-stats table of a rocprofv3 run (rocpd sqlite db or csv dir) as text for profiles/."""
import glob
import sqlite3
import sys

path = sys.argv[1]
dbs = glob.glob(path + "/**/*.db", recursive=True)
if dbs:
    c = sqlite3.connect(dbs[0])
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print(f"# source: {dbs[0]}  (durations in ns)")
    print(f"{'calls':>7} {'total_ms':>11} {'avg_us':>11} {'%':>7}  kernel")
    for name, calls, total, avg, pct in rows:
        # top_kernels reports microseconds on this rocprofv3 build
        print(f"{calls:7d} {total / 1e3:11.3f} {avg:11.3f} {pct:7.2f}  {name[:150]}")
else:
    print("no rocpd database under", path)
