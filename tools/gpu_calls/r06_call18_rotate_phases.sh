#!/bin/bash
# Round 6, call 18: tree_rotate.h's phases inside an adaptation (config 4 and 5, waiting mode).
O=gpurun_out/r06_call18; mkdir -p $O
P="import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
print(d['value'], d['adaptation'].get('seconds_to_adapted'), [l for l in d['cold_job'].get('trees', []) + d['config'].get('trees', []) if l.startswith('adaptive')])"
A="--steps 2 --warmup 1 --no-cpu-baseline --per-frame-frames 0 --surface-area-fold-steps 0 --cold-job-spp 16"
for cfg in 4 5; do
  timeout 600 python bench.py --config $cfg $A > $O/bench_cfg${cfg}.json 2>> $O/bench.err; echo cfg $cfg: $(python -c "$P" $O/bench_cfg${cfg}.json 2>&1 | tail -1)
  python -c "
import json; d=json.loads(open('$O/bench_cfg${cfg}.json').read().strip().split('\n')[-1]); print(json.dumps(d['adaptation'])[:600]); print(d['config'].keys())"
done
grep -v amdgpu.ids $O/bench.err | tail -5
