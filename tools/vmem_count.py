"""Per kernel: launches, average duration, vector-memory wave-instructions per launch (SQ_INSTS_VMEM; the _RD / _WR pair crashes rocprofv3 7.2 on this image) and what they would
cost at 30 clocks per instruction and CU (256 CUs, 2.4 GHz) as a fraction of the duration.  usage: python tools/vmem_count.py <rocprofv3 dir>"""
import glob, sqlite3, sys, collections
for db in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    con = sqlite3.connect(db)
    cnt = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for k, c, v, m in con.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"):
        cnt[k][c] += v
        n[k] = max(n[k], m)
    dur = {}
    try:
        for k, calls, total, avg, pct in con.execute("select * from top_kernels"):
            dur[k] = avg
    except Exception:
        objs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
        for obj in objs:
            cols = [c[1] for c in con.execute("pragma table_info('%s')" % obj)]
            if {"name", "start", "end"} <= set(cols):
                acc = collections.defaultdict(list)
                for name, s, e in con.execute("select name, start, end from '%s'" % obj):
                    acc[name].append((e - s) / 1e3)
                dur = {k: sum(v) / len(v) for k, v in acc.items()}
                break
    print("%-60s %7s %10s %12s %12s %8s" % ("kernel", "calls", "avg_us", "vmem(_rd)", "vmem_wr", "x30clk"))
    rows = []
    for k in cnt:
        rd = (cnt[k].get("SQ_INSTS_VMEM_RD", 0) + cnt[k].get("SQ_INSTS_VMEM", 0)) / max(n[k], 1)       # SQ_INSTS_VMEM: loads + stores
        wr = cnt[k].get("SQ_INSTS_VMEM_WR", 0) / max(n[k], 1)
        d = dur.get(k, 0.0)
        est = (rd + wr) / 256.0 * 30.0 / 2400.0          # us
        rows.append((d * n[k], k, n[k], d, rd, wr, est / d if d else 0.0))
    for _, k, m, d, rd, wr, fr in sorted(rows, reverse=True)[:24]:
        print("%-60s %7d %10.1f %12.0f %12.0f %8.2f" % (k.split("(")[0].replace("pips::", "").replace("void ", "")[:60], m, d, rd, wr, fr))
