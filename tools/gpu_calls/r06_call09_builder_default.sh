#!/bin/bash
# Round 6, call 9: RT_CTX_OPT_TREE_BUILDER = 2 as the default (the device-built candidate first, the host's build abandoned when it wins): the suite, the driver's command,
# configs 2 / 3 / 5 / 1 (where the host's candidate still wins or nobody does: nothing may get slower than call 5's lines).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_call09
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 1200 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/pytest_gpu.log 2>&1; el suite: $(grep -aE "passed|failed|rror" $O/pytest_gpu.log | tail -1)
grep -aE "^E  |^FAILED" $O/pytest_gpu.log | head -10
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; el smoke: $(tail -1 $O/smoke.log)
RT_FUZZ_SEEDS=1000 timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -n 32 -p no:cacheprovider > $O/fuzz_1000_seeds.log 2>&1; el fuzz: $(tail -1 $O/fuzz_1000_seeds.log)
( time python bench.py ) > $O/bench_driver_command.json 2> $O/bench.err; el bench: $(python -c "
import json; d=json.loads(open('$O/bench_driver_command.json').read().strip().splitlines()[-1]); p=d['per_frame']; a=p.get('samples_ahead') or {}; r=d['roofline']; c=d['config']
print(d['value'], d['ms_per_step'], 'per frame', p['ms_per_frame'], a.get('bit_identical_to_rt_integrate_of_the_same_samples'), 'moving', p['moving_camera']['with_over_without'], 'parity', d['parity']['bit_identical'], 'roofline', r.get('frac'), r.get('stale'), 'cold', {k: v for k, v in (d['cold_job'] or {}).items() if k not in ('what', 'trees')}, 'setup', c.get('setup_s'), c.get('setup_breakdown'), c.get('path_state_alloc_s'), 'adapt', d['adaptation'].get('seconds_to_adapted'), [t for t in c.get('trees', []) if t.startswith('upload') or t.startswith('shadow')])" 2>&1 | tail -1)
grep real $O/bench.err
for cfg in 2 3 5 1; do
  extra="--no-cpu-baseline"; [ $cfg = 1 ] && extra="--no-cpu-baseline --steps 64 --warmup 4 --per-frame-frames 192"
  timeout 600 python bench.py --config $cfg $extra > $O/bench_cfg$cfg.json 2>> $O/bench.err; el cfg $cfg: $(python -c "
import json; d=json.loads(open('$O/bench_cfg$cfg.json').read().strip().splitlines()[-1]); f=d['per_frame']; a=f.get('samples_ahead') or {}; c=d['config']
print(d['value'], 'per frame', f['ms_per_frame'], a.get('bit_identical_to_rt_integrate_of_the_same_samples'), 'sa fold', (d['surface_area_fold'] or {}).get('value'), 'cold', {k: v for k, v in (d['cold_job'] or {}).items() if k not in ('what', 'trees')}, c.get('setup_breakdown'), [t for t in c.get('trees', []) if t.startswith('upload') or t.startswith('shadow')])" 2>&1 | tail -1)
done
el all done
