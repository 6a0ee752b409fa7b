#!/bin/bash
# Round 6, call 17: where an adaptation's worker spends its time (the report line's new stage list), configs 4 and 5, waiting (27: device folds) and asynchronous (25: host folds).
O=gpurun_out/r06_call17; mkdir -p $O
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
P="import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
print(d['value'], d['adaptation'].get('seconds_to_adapted'), [l for l in d['config'].get('trees', []) if l.startswith('adaptive')])"
A="--steps 2 --warmup 1 --no-cpu-baseline --per-frame-frames 0 --surface-area-fold-steps 0 --cold-job-spp 0"
for cfg in 4 5; do for mode in 27 25; do
  timeout 600 python bench.py --config $cfg $A --adaptive-fold $mode > $O/bench_cfg${cfg}_mode$mode.json 2>> $O/bench.err; el cfg $cfg mode $mode: $(python -c "$P" $O/bench_cfg${cfg}_mode$mode.json 2>&1 | tail -1)
done; done
grep -v amdgpu.ids $O/bench.err | tail -5
