#!/bin/bash
# Round 4, call 10: per-frame pattern, the instance with loop D for every launch (RT_OPT_TRACE_TAIL_PATHS large) while the in-kernel
# chunk / refill decision moves (RT_OPT_SMALL_LAUNCH_PATHS): do the 1 - 2 M-ray launches of a frame run better refilled now that
# the refill tail has loop D too?  And the headline with the loop-D instance everywhere (expected: -2 %).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_call10
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
for sl in 3000000 1500000 1000000 500000 200000 0; do
  python bench.py --steps 1 --no-cpu-baseline --per-frame-frames 96 --per-frame-only --tail-paths 100000000 --small-launch-paths $sl > $O/pf_small$sl.json 2>> $O/bench.err; el small-launch $sl: $(python -c "
import json; d=json.loads(open('$O/pf_small$sl.json').read().strip().splitlines()[-1]); print(d['per_frame']['mrays_per_s'], d['per_frame']['ms_per_frame'])")
done
for cfg in 2 3; do for sl in 3000000 1000000; do
  python bench.py --config $cfg --steps 1 --no-cpu-baseline --per-frame-frames 96 --per-frame-only --tail-paths 100000000 --small-launch-paths $sl > $O/pf_cfg${cfg}_small$sl.json 2>> $O/bench.err; el cfg $cfg small-launch $sl: $(python -c "
import json; d=json.loads(open('$O/pf_cfg${cfg}_small$sl.json').read().strip().splitlines()[-1]); print(d['per_frame']['mrays_per_s'], d['per_frame']['ms_per_frame'])")
done; done
python bench.py --steps 3 --no-cpu-baseline --per-frame-frames 0 --tail-paths 4000000000 > $O/bench_tail_everywhere.json 2>> $O/bench.err; el headline, loop-D instance everywhere: $(python -c "
import json; d=json.loads(open('$O/bench_tail_everywhere.json').read().strip().splitlines()[-1]); print(d['value'], (d['roofline'].get('live_isolated') or {}).get('kernel_ms_per_spp'))")
python bench.py --samples-in-flight 8 --steps 8 --samples-per-step 8 --no-cpu-baseline --per-frame-frames 0 > $O/bench_8_in_flight.json 2>> $O/bench.err; el 8 in flight: $(python -c "
import json; d=json.loads(open('$O/bench_8_in_flight.json').read().strip().splitlines()[-1]); print(d['value'])")
python bench.py --samples-in-flight 8 --steps 8 --samples-per-step 8 --no-cpu-baseline --per-frame-frames 0 --tail-paths 4000000000 > $O/bench_8_in_flight_tail.json 2>> $O/bench.err; el 8 in flight, loop D: $(python -c "
import json; d=json.loads(open('$O/bench_8_in_flight_tail.json').read().strip().splitlines()[-1]); print(d['value'])")
tail -3 $O/bench.err | grep -v amdgpu.ids
el all done
