#!/bin/bash
# Round 6, call 22: kernel traces of the N = 8 tile's 256-spp job and of the whole frame's, to see which launches carry the tile's fixed cost.
O=$PWD/gpurun_out/r06_call22; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp
for t in 8 1; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_tiles$t -o t -- python $R/tools/tile_job_trace.py --tiles $t --rank 0 --spp 256 --repeat 3 > $O/tiles$t.log 2>&1
  tail -1 $O/tiles$t.log
  f=$(find $O/trace_tiles$t -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_tiles$t.csv
  k=$(find $O/trace_tiles$t -name "*kernel_trace.csv" | head -1); python - $k $O/launches_tiles$t.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
with open(sys.argv[2], "w") as f:
    t0 = int(rows[0]["Start_Timestamp"])
    for r in rows:
        f.write("%s,%d,%d\n" % (r["Kernel_Name"][:60].replace(",", ";"), int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
PY
  rm -rf $O/trace_tiles$t
done
python - $O <<'PY'
import csv, sys, os
O = sys.argv[1]
def load(t):
    d = {}
    for r in csv.DictReader(open(os.path.join(O, "kernel_stats_tiles%d.csv" % t))):
        d[r["Name"][:70]] = (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6)
    return d
a, b = load(8), load(1)
print("%-72s %8s %10s %8s %10s %8s" % ("kernel", "calls/8", "ms(tile)", "calls/1", "ms(full)/8", "ratio"))
for k in sorted(b, key=lambda k: -b[k][1]):
    if k in a: print("%-72s %8d %10.3f %8d %10.3f %8.3f" % (k, a[k][0], a[k][1], b[k][0], b[k][1] / 8.0, a[k][1] / (b[k][1] / 8.0)))
print("sum tile %.2f ms, sum full / 8 %.2f ms" % (sum(v[1] for v in a.values()), sum(v[1] for v in b.values()) / 8.0))
PY
