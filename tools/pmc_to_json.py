"""FETCH_SIZE / WRITE_SIZE passes (rocprofv3 --pmc, rocpd sqlite output) -> profiles/rN_pmc_traffic.json.
usage: python tools/pmc_to_json.py out.json <fetch_dir> <write_dir> [<fetch_dir2> <write_dir2> ...]
(kernels already present from an earlier pair are kept: list the headline passes first)
FETCH_SIZE / WRITE_SIZE are in KiB-sized units of 1024 B?  No: rocprofv3 reports them in KB (1e3?) -- we keep the
tool's own unit factor 1024 (FETCH_SIZE = TCC_EA0_RDREQ x 64 B / 1024, MI355X_MICROARCH.md) and apply the gfx950
correction the guide prescribes for 16-byte-per-lane streaming reads: hbm_bytes = 2 x FETCH + WRITE."""
import glob, json, sqlite3, sys

def per_kernel(d, counter):
    out = {}
    for db in glob.glob(d + "/**/*.db", recursive=True):
        cur = sqlite3.connect(db).cursor()
        for k, v, n in cur.execute("select kernel_name, avg(value), count(*) from counters_collection where counter_name = ? "
                                   "group by kernel_name", (counter,)):
            out[k.split("(")[0].replace("void ", "")] = (v, n)
    return out

res = {}
args = sys.argv[2:]
for i in range(0, len(args), 2):
    f, w = per_kernel(args[i], "FETCH_SIZE"), per_kernel(args[i + 1], "WRITE_SIZE")
    for k in f:
        if not k.startswith("pips::") or k in res:
            continue
        fb, wb = f[k][0] * 1024.0, w.get(k, (0.0, 0))[0] * 1024.0
        res[k] = {"launches": f[k][1], "FETCH_SIZE_bytes": fb, "WRITE_SIZE_bytes": wb, "hbm_bytes": 2.0 * fb + wb}
json.dump({"note": "per-launch averages; hbm_bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE tallies the 128-B requests "
                   "of 16-B/lane streaming reads at 64 B, MI355X_MICROARCH.md); separate --pmc passes of the commands in "
                   "profiles/README.md", "kernels": res}, open(sys.argv[1], "w"), indent=1)
print("wrote", sys.argv[1], len(res), "kernels")
