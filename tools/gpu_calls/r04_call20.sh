#!/bin/bash
# Round 4, call 20: with lanes refilled from a wave's own chunks, up to which launch size does STATIC assignment beat the shared work
# heads?  Batches of 4 / 8 / 16 samples of the 1080p frame in flight (the TAIL instance), RT_OPT_SMALL_LAUNCH_PATHS 3 M .. always.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_call20
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
v() { python -c "
import json; d=json.loads(open('$O/$1.json').read().strip().splitlines()[-1]); print(d['value'])"; }
for s in 4 8 16; do for sl in 3000000 8000000 20000000 4000000000; do
  python bench.py --samples-in-flight $s --steps 8 --samples-per-step $s --no-cpu-baseline --per-frame-frames 0 --small-launch-paths $sl > $O/b_${s}_sl$sl.json 2>> $O/bench.err; el $s in flight, chunk mode below $sl rays: $(v b_${s}_sl$sl)
done; done
python bench.py --path-state-gb 16 --steps 3 --no-cpu-baseline --per-frame-frames 0 --small-launch-paths 20000000 > $O/b_16g_sl20.json 2>> $O/bench.err; el 16 GiB, below 20 M: $(v b_16g_sl20)
grep -v amdgpu.ids $O/bench.err | tail -3
el all done
