#!/bin/bash
# Round 4, call 6: rt_frame_present (the per-frame pattern's image travels while the next frame is traced): suite, per-frame
# legs of configs 4 / 2 / 3, a kernel trace of a few frames as a timeline (where are the gaps?), and the libm tolerance series
# again with the resolved-image column and the heavy-tail diagnostics.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_call06
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 900 python -m pytest tests -q -m gpu -x -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror|FAILED" | tail -5 > $O/pytest_gpu.log; el suite: $(tail -1 $O/pytest_gpu.log)
for cfg in 4 2 3; do
  python bench.py --config $cfg --steps 1 --no-cpu-baseline --per-frame-frames 96 --per-frame-only > $O/pf_cfg$cfg.json 2>> $O/bench.err; el per-frame cfg $cfg: $(python -c "
import json; d=json.loads(open('$O/pf_cfg$cfg.json').read().strip().splitlines()[-1]); print(d['per_frame']['mrays_per_s'], d['per_frame']['ms_per_frame'])")
done
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/bench.py --steps 1 --no-cpu-baseline --per-frame-frames 6 --per-frame-only > $O/trace_run.log 2>&1 )
CSV=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python - <<PY > $O/per_frame_gantt.log
import csv
rows = list(csv.DictReader(open("$CSV")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last full frame: from the last k_raygen to the k_resolve after it
idx = [i for i, r in enumerate(rows) if "k_raygen" in r["Kernel_Name"]]
a = idx[-2]
b = next(i for i in range(a, len(rows)) if "k_resolve" in rows[i]["Kernel_Name"])
t0 = int(rows[a]["Start_Timestamp"])
prev_end = {}
busy_until = 0
gap_total = 0
print("one frame of the per-frame pattern: start us, end us, duration us, queue, kernel (grid) | idle gap before it when nothing else ran")
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]
    gap = max(0, s - busy_until) if busy_until else 0
    gap_total += gap
    print("%9.1f %9.1f %8.1f  q%-3s %-44s grid %s | %.1f" % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), name, r.get("Grid_Size", "?"), gap / 1e3))
    busy_until = max(busy_until, e)
print("frame span %.1f us, of which NO kernel was running for %.1f us" % ((int(rows[b]["End_Timestamp"]) - t0) / 1e3, gap_total / 1e3))
PY
tail -3 $O/per_frame_gantt.log
rm -rf $O/trace
( timeout 1200 python -X faulthandler tools/libm_tolerance_series.py > $O/libm_tolerance_series_cfg5.json 2> $O/libm_series.err ); el series: $(python -c "
import json; d = json.load(open('$O/libm_tolerance_series_cfg5.json')); print([(p['spp'], '%.2e' % p['rel_l2'], '%.2e' % p['rel_l2_resolved'], p['share_of_worst_pixel'], p['nan_pixels'], p['nan_positions_equal']) for p in d['series']], d['fitted_slope'], d['resolved'])")
grep -v amdgpu.ids $O/libm_series.err | tail -3
tail -3 $O/bench.err | grep -v amdgpu.ids
el all done
