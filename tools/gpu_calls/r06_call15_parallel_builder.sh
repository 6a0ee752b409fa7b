#!/bin/bash
# Round 6, call 15: own_bvh.h's top levels on the pool -- configs 5 and 2 (where rt_scene_upload waits for the host-built candidate), the suite, the driver's command.
O=gpurun_out/r06_call15; mkdir -p $O
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
P="import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); c=d.get('cold_job') or {}; pf=d.get('per_frame') or {}
print(d['value'], d['ms_per_step'], 'per frame', pf.get('ms_per_frame'), 'parity', (d.get('parity') or {}).get('bit_identical'), 'cold', {k: v for k, v in c.items() if k not in ('what', 'trees', 'full_batch')}, 'setup', d['config'].get('setup_s'), d['config'].get('scene_s'), d['config'].get('setup_breakdown'), d['config'].get('path_state_alloc_s'))
print([l for l in c.get('trees', []) if l.startswith('upload')])"
for cfg in 5 2; do
  timeout 900 python bench.py --config $cfg --no-cpu-baseline --per-frame-frames 0 --surface-area-fold-steps 0 > $O/bench_cfg$cfg.json 2>> $O/bench.err; el cfg $cfg: $(python -c "$P" $O/bench_cfg$cfg.json 2>&1 | tail -2)
done
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; el suite: $(tail -1 $O/pytest_gpu.log)
( time timeout 900 python bench.py > $O/bench_driver_command.json 2>> $O/bench.err ) 2>&1 | grep real; el bench: $(python -c "$P" $O/bench_driver_command.json 2>&1 | tail -2)
grep -v amdgpu.ids $O/bench.err | tail -5
