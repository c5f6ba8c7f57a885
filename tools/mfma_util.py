"""MFMA utilisation per kernel from a rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES run (csv output):
busy cycles summed over the 1024 SIMDs / (1024 * kernel duration * shader clock).
usage: python tools/mfma_util.py <dir with *_counter_collection.csv and *_kernel_trace.csv> [clock_GHz]"""
import collections, csv, glob, os, sys
d = sys.argv[1]
ghz = float(sys.argv[2]) if len(sys.argv) > 2 else 2.4
cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in csv.DictReader(open(cc)):
    if r["Counter_Name"] != "SQ_VALU_MFMA_BUSY_CYCLES":
        continue
    name, ns = dur[r["Dispatch_Id"]]
    a = acc[name[:96]]
    a[0] += float(r["Counter_Value"]); a[1] += ns; a[2] += 1
print("%-98s %6s %10s %10s" % ("kernel", "calls", "avg_us", "mfma_util"))
for k, (busy, ns, n) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    if busy > 0:
        print("%-98s %6d %10.2f %10.3f" % (k, n, ns / n / 1e3, busy / (1024.0 * ns * ghz)))
