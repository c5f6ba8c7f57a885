"""Mean per-launch PMC values of the kernels matching a substring (rocprofv3 --pmc ... --output-format csv).
usage: python tools/pmc_summary.py <counter_collection.csv> <kernel-substring>"""
import collections, csv, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in row["Kernel_Name"]:
        acc[row["Kernel_Name"][:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in acc.items():
    print(k)
    for c, vals in sorted(v.items()):
        print("   %-30s mean %.4g  (n=%d)" % (c, sum(vals) / len(vals), len(vals)))
