#!/bin/bash
# Round 6, call 46: the last check of the final tree: the suite, smoke(), the driver's bench command, 2000 host-layer walks.
O=gpurun_out/r06_call46; mkdir -p $O
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/pytest_gpu.log 2>&1; el suite: $(grep -aE "passed|failed" $O/pytest_gpu.log | tail -1)
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; el smoke: $(tail -1 $O/smoke.log)
( time python bench.py ) > $O/bench_driver_command.json 2> $O/bench.err; el bench: $(python -c "
import json; d=json.loads(open('$O/bench_driver_command.json').read().strip().splitlines()[-1]); p=d['per_frame']; a=p.get('samples_ahead') or {}; r=d['roofline']; c=d['config']
print(d['value'], d['ms_per_step'], 'per frame', p['ms_per_frame'], a.get('bit_identical_to_rt_integrate_of_the_same_samples'), 'moving', p['moving_camera']['ms_per_frame'], p['moving_camera']['with_over_without'], 'parity', d['parity']['bit_identical'], 'roofline', r.get('frac'), r.get('stale'), 'cold', {k: v for k, v in (d['cold_job'] or {}).items() if k in ('upload_s', 'render_s', 'wall_s', 'non_finite_pixels', 'over_the_warm_headline')}, 'setup', c.get('setup_s'), 'adapt', d['adaptation'].get('seconds_to_adapted'), 'cpu', d['cpu_baseline']['value'])" 2>&1 | tail -1)
grep real $O/bench.err
RT_HOST_SEQ_SEEDS=2000 timeout 1200 python -m pytest tests/test_gpu_samples_ahead.py -k through_the_integrator_equal -q -m gpu -n 8 -p no:cacheprovider > $O/host_walks_2000.log 2>&1; el host walks: $(tail -1 $O/host_walks_2000.log)
