#!/bin/bash
# Round 4, call 22: the crossover of the loop-D instance again, now that its chunks refill and its chunk threshold is 8 M rays:
# 32 / 64 / 128 samples in flight with the plain instance (default) and with the loop-D instance (RT_OPT_TRACE_TAIL_PATHS huge).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_call22
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
v() { python -c "
import json; d=json.loads(open('$O/$1.json').read().strip().splitlines()[-1]); print(d['value'])"; }
for s in 32 64 128; do
  python bench.py --samples-in-flight $s --steps 3 --samples-per-step $s --no-cpu-baseline --per-frame-frames 0 > $O/b_$s.json 2>> $O/bench.err
  python bench.py --samples-in-flight $s --steps 3 --samples-per-step $s --no-cpu-baseline --per-frame-frames 0 --tail-paths 4000000000 > $O/b_${s}_tail.json 2>> $O/bench.err
  el $s in flight: plain $(v b_$s) loop-D instance $(v b_${s}_tail)
done
el all done
