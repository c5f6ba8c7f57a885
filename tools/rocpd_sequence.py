"""Print the launch sequence (kernel, duration in us) of ONE forward out of a rocprofv3 --kernel-trace rocpd database: the launches
from the last stem convolution up to the first gather after it = the encoder, layer by layer (the per-kernel summary of
rocpd_summary.py lumps every layer that shares a template instantiation).
usage: python tools/rocpd_sequence.py <results.db> [out.txt] [first-kernel-substring=stem_conv] [stop-substring=mixer_input]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
start_key = sys.argv[3] if len(sys.argv) > 3 else "stem_conv"
stop_key = sys.argv[4] if len(sys.argv) > 4 else "mixer_input"
rows = None
objs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
for obj in sorted(objs, key=lambda n: (0 if n == "kernels" else 1, n)):
    cols = [c[1] for c in db.execute("pragma table_info('%s')" % obj)]
    if {"name", "start", "end"} <= set(cols):
        rows = list(db.execute("select name, start, end from '%s' order by start" % obj))
        break
if rows is None:
    print("no table with (name, start, end); objects:", objs)
    sys.exit(1)
stems = [i for i, r in enumerate(rows) if start_key in r[0]]
i0 = stems[-1]
out = []
t_prev_end = None
for name, s, e in rows[i0:]:
    if stop_key in name and len(out) > 1:
        break
    short = re.sub(r"\(.*", "", name).replace("pips::", "").replace("void ", "")
    gap = 0.0 if t_prev_end is None else (s - t_prev_end) / 1e3
    out.append("%-78s %9.1f us   (gap %5.1f)" % (short[:78], (e - s) / 1e3, gap))
    t_prev_end = e
txt = "\n".join(out)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
