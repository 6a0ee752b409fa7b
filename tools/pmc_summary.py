"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (mean per dispatch).
  python tools/pmc_summary.py <dir>          trace / shade kernels only, mean per dispatch
  python tools/pmc_summary.py <dir> --all    every kernel, one line per dispatch (micro-benchmarks)"""
import csv, glob, sys, collections
d = sys.argv[1]
every = "--all" in sys.argv
for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
    print("==", f.split("/")[-1])
    if every:
        rows = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            key = (int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0].replace("void ", ""))
            rows.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
        for (did, k), cs in sorted(rows.items()):
            print("   #%d %s %s" % (did, k, {c: "%.5g" % v for c, v in cs.items()}))
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_trace" not in k and "k_shade" not in k and "k_flush" not in k and "k_raygen" not in k:
            continue
        short = k.split("(")[0].replace("void ", "")
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print("  ", k, {c: "%.4g (n=%d)" % (sum(v) / len(v), len(v)) for c, v in cs.items()})
