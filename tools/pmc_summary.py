"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (mean per dispatch)."""
import csv, glob, sys, collections
d = sys.argv[1]
for f in sorted(glob.glob(d + "/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_trace" not in k and "k_shade" not in k:
            continue
        short = k.split("(")[0].replace("void ", "")
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", f.split("/")[-1])
    for k, cs in acc.items():
        print("  ", k, {c: "%.4g (n=%d)" % (sum(v) / len(v), len(v)) for c, v in cs.items()})
