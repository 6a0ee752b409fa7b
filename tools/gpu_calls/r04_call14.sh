#!/bin/bash
# Round 4, call 14: camera-ray launches refilled instead of chunked (RT_OPT_FIRST_BOUNCE_REFILL, a host-side launch parameter:
# the code object is the one of call 13) -- the per-frame pattern on configs 4 / 2 / 3 / 5, and small batches.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_call14
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
pf() { python -c "
import json; d=json.loads(open('$O/$1.json').read().strip().splitlines()[-1]); print(d['per_frame']['mrays_per_s'], d['per_frame']['ms_per_frame'])"; }
for cfg in 4 2 3 5; do
  for r in 0 1 500000; do
    python bench.py --config $cfg --steps 1 --no-cpu-baseline --per-frame-frames 64 --per-frame-only --first-bounce-refill $r > $O/pf_cfg${cfg}_refill$r.json 2>> $O/bench.err; el cfg $cfg first-bounce refill $r: $(pf pf_cfg${cfg}_refill$r)
  done
done
for s in 1 2 4; do for r in 0 1; do
  python bench.py --samples-in-flight $s --steps 16 --samples-per-step $s --no-cpu-baseline --per-frame-frames 0 --first-bounce-refill $r > $O/b_${s}_refill$r.json 2>> $O/bench.err; el $s in flight, refill $r: $(python -c "
import json; d=json.loads(open('$O/b_${s}_refill$r.json').read().strip().splitlines()[-1]); print(d['value'])")
done; done
grep -v amdgpu.ids $O/bench.err | tail -3
el all done
