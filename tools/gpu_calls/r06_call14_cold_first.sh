#!/bin/bash
# Round 6, call 14: the bench with its cold_job leg moved to the front of the process (nothing released before it), the driver's command and configs 2, 3, 5.
O=gpurun_out/r06_call14; mkdir -p $O
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
P="import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); c=d['cold_job'] or {}
print(d['value'], d['ms_per_step'], 'per frame', d['per_frame']['ms_per_frame'], 'parity', d['parity']['bit_identical'], 'cold', {k: v for k, v in c.items() if k not in ('what', 'trees')}, 'setup', d['config'].get('setup_s'), d['config'].get('scene_s'), d['config'].get('setup_breakdown'), d['config'].get('path_state_alloc_s'))"
( time timeout 900 python bench.py > $O/bench_driver_command.json 2> $O/bench.err ) 2>&1 | grep real; el bench: $(python -c "$P" $O/bench_driver_command.json 2>&1 | tail -1)
for cfg in 2 3 5; do
  timeout 900 python bench.py --config $cfg --no-cpu-baseline --per-frame-frames 0 --surface-area-fold-steps 0 > $O/bench_cfg$cfg.json 2>> $O/bench.err; el cfg $cfg: $(python -c "$P" $O/bench_cfg$cfg.json 2>&1 | tail -1)
done
tail -5 $O/bench.err
