"""Prints the kernel launches of a rocprofv3 --kernel-trace CSV as a timeline (start, end, duration in us
relative to the first listed launch, short kernel name, queue), for a window of launches."""
import csv, sys
path, first, count = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[first]["Start_Timestamp"])
for r in rows[first:first + count]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"]
    short = name.split("(")[0].replace("void ", "")[:40]
    print("%10.1f %10.1f %9.1f  q%-3s %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), short))
