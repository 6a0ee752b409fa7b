#!/bin/bash
# Round 6, call 25: the host's fold on a pool -- the suite (device folds are compared with the host's record for record there), the asynchronous adaptation's time
# on configs 4 / 2 / 5, the driver's command.
O=gpurun_out/r06_call25; mkdir -p $O
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
P="import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); c=d.get('cold_job') or {}; pf=d.get('per_frame') or {}; mc=pf.get('moving_camera') or {}
print(d['value'], d['ms_per_step'], 'adapted in', d['adaptation'].get('seconds_to_adapted'), 'per frame', pf.get('ms_per_frame'), 'moving', mc.get('ms_per_frame'), mc.get('with_over_without'), 'parity', (d.get('parity') or {}).get('bit_identical'), 'cold', {k: v for k, v in c.items() if k in ('upload_s', 'render_s', 'wall_s', 'over_the_warm_headline')}, 'setup', d['config'].get('setup_s'), d['config'].get('setup_breakdown'), 'sa fold', (d.get('surface_area_fold') or {}).get('value'))"
for c in 4 2 5; do timeout 300 python tools/async_adaptation_time.py --config $c 2>&1 | tail -1 | cut -c1-520; done
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; el suite: $(grep -E "passed|failed" $O/pytest_gpu.log | tail -1)
( time timeout 900 python bench.py > $O/bench_driver_command.json 2>> $O/bench.err ) 2>&1 | grep real; el bench: $(python -c "$P" $O/bench_driver_command.json 2>&1 | tail -1)
timeout 900 python bench.py --device-fold 0 --steps 2 --warmup 1 --no-cpu-baseline --per-frame-frames 0 --surface-area-fold-steps 0 --cold-job-spp 0 > $O/bench_host_folds.json 2>> $O/bench.err; el host folds only: $(python -c "$P" $O/bench_host_folds.json 2>&1 | tail -1)
grep -v amdgpu.ids $O/bench.err | tail -5
