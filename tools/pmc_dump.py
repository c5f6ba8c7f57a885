"""Print per-kernel averages of the counters in a rocprofv3 results db.  usage: python tools/pmc_dump.py <dir> [kernel-substring]"""
import sys, glob, sqlite3
pat = sys.argv[2] if len(sys.argv) > 2 else "gather_tiled"
for db in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    con = sqlite3.connect(db)
    cur = con.cursor()
    try:
        cols = [d[0] for d in cur.execute("select * from counters_collection limit 1").description]
    except Exception as e:
        print(db, "no counters_collection", e); continue
    rows = cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
    for k, c, v, n in rows:
        if pat in k:
            print(f"{k.split('(')[0][:40]:40s} {c:28s} avg {v:16.1f}  (n={n})")
