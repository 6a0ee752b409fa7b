#!/bin/bash
# Round 4, call 23: per-frame pattern, persistent-grid size again now that chunks refill (RT_OPT_TRACE_WAVES_PER_CU 26 .. 10), and the
# loop thresholds (node_q : leaf_q) of the refilled chunk mode.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_call23
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
pf() { python -c "
import json; d=json.loads(open('$O/$1.json').read().strip().splitlines()[-1]); print(d['per_frame']['mrays_per_s'], d['per_frame']['ms_per_frame'])"; }
for w in 26 22 18 14 10; do
  python bench.py --steps 1 --no-cpu-baseline --per-frame-frames 64 --per-frame-only --trace-waves $w > $O/pf_w$w.json 2>> $O/bench.err; el trace waves per CU $w: $(pf pf_w$w)
done
for t in 0x0820 0x0818 0x1020 0x0430 0x0810; do
  python bench.py --steps 1 --no-cpu-baseline --per-frame-frames 64 --per-frame-only --trace-tune $t > $O/pf_t$t.json 2>> $O/bench.err; el trace tune $t: $(pf pf_t$t)
done
el all done
