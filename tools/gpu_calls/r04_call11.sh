#!/bin/bash
# Round 4, call 11: where does the loop-D instance stop paying?  Batches of 16 / 32 / 64 samples of the 1080p frame in flight with the
# plain instance (RT_OPT_TRACE_TAIL_PATHS at its default 3 M) and with the loop-D instance (huge); 32 / 16 GiB budgets likewise.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_call11
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
v() { python -c "
import json; d=json.loads(open('$O/$1.json').read().strip().splitlines()[-1]); print(d['value'])"; }
for s in 16 32 64; do
  python bench.py --samples-in-flight $s --steps 4 --samples-per-step $s --no-cpu-baseline --per-frame-frames 0 > $O/b_${s}.json 2>> $O/bench.err
  python bench.py --samples-in-flight $s --steps 4 --samples-per-step $s --no-cpu-baseline --per-frame-frames 0 --tail-paths 4000000000 > $O/b_${s}_tail.json 2>> $O/bench.err
  el $s in flight: plain $(v b_${s}) loop-D instance $(v b_${s}_tail)
done
for g in 32 16; do
  python bench.py --path-state-gb $g --steps 3 --no-cpu-baseline --per-frame-frames 0 > $O/b_${g}GiB.json 2>> $O/bench.err
  python bench.py --path-state-gb $g --steps 3 --no-cpu-baseline --per-frame-frames 0 --tail-paths 4000000000 > $O/b_${g}GiB_tail.json 2>> $O/bench.err
  el $g GiB: plain $(v b_${g}GiB) loop-D instance $(v b_${g}GiB_tail)
done
tail -3 $O/bench.err | grep -v amdgpu.ids
el all done
