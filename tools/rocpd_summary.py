"""Dump the per-kernel summary (rocprofv3 --kernel-trace --stats, rocpd sqlite output) as text.
usage: python tools/rocpd_summary.py <results.db> [out.txt]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select * from top_kernels"))
lines = ["%-110s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct")]
for name, calls, total, avg, pct in rows:
    lines.append("%-110s %8d %14.1f %12.2f %7.2f" % (name[:110], calls, total / 1e0 if total > 1e6 else total, avg, pct))
txt = "\n".join(lines)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
