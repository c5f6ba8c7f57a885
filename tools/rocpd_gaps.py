"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace rocpd database: total busy, total gap, gap histogram, per forward.
usage: python tools/rocpd_gaps.py <results.db> [first-kernel-substring=stem_conv]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
key = sys.argv[2] if len(sys.argv) > 2 else "stem_conv"
rows = None
objs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
for obj in sorted(objs, key=lambda n: (0 if n == "kernels" else 1, n)):
    cols = [c[1] for c in db.execute("pragma table_info('%s')" % obj)]
    if {"name", "start", "end"} <= set(cols):
        rows = list(db.execute("select name, start, end from '%s' order by start" % obj))
        break
stems = [i for i, r in enumerate(rows) if key in r[0]]
i0, i1 = stems[-6], stems[-1]                       # five whole forwards
sel = rows[i0:i1]
busy = sum(e - s for _, s, e in sel) / 1e3
span = (sel[-1][2] - sel[0][1]) / 1e3
gaps = [(sel[k + 1][1] - sel[k][2]) / 1e3 for k in range(len(sel) - 1)]
n = len(sel) / 5.0
print("5 forwards: %.0f kernels each, span %.3f ms, busy %.3f ms, idle %.3f ms per forward" % (n, span / 5e3, busy / 5e3, (span - busy) / 5e3))
import collections
h = collections.Counter(min(int(g), 10) for g in gaps)
print("gap histogram (us -> launches per forward):", {k: round(v / 5.0, 1) for k, v in sorted(h.items())})
big = sorted(((g, sel[k][0][:60], sel[k + 1][0][:60]) for k, g in enumerate(gaps)), reverse=True)[:8]
for g, a, b in big:
    print("  %.1f us between %s -> %s" % (g, a, b))
