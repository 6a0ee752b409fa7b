#!/bin/bash
# Round 6, call 19: tree rotations on the pool -- an adaptation's stages on configs 4 and 5 (waiting mode), then the driver's command (moving-camera leg: the asynchronous worker beside frames).
O=gpurun_out/r06_call19; mkdir -p $O
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
P="import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); pf=d.get('per_frame') or {}; mc=pf.get('moving_camera') or {}
print(d['value'], 'adapted in', d['adaptation'].get('seconds_to_adapted'), 'per frame', pf.get('ms_per_frame'), 'moving', mc.get('ms_per_frame'), mc.get('with_over_without'), 'parity', (d.get('parity') or {}).get('bit_identical'))
print([l[l.find('on a worker thread') - 6:][:420] for l in d['config'].get('trees', []) if 'adaptive fold' in l])"
A="--steps 2 --warmup 1 --no-cpu-baseline --per-frame-frames 0 --surface-area-fold-steps 0 --cold-job-spp 0"
for cfg in 4 5; do
  timeout 600 python bench.py --config $cfg $A > $O/bench_cfg${cfg}.json 2>> $O/bench.err; el cfg $cfg: $(python -c "$P" $O/bench_cfg${cfg}.json 2>&1 | tail -2)
done
( time timeout 900 python bench.py > $O/bench_driver_command.json 2>> $O/bench.err ) 2>&1 | grep real; el bench: $(python -c "$P" $O/bench_driver_command.json 2>&1 | tail -2)
grep -v amdgpu.ids $O/bench.err | tail -5
