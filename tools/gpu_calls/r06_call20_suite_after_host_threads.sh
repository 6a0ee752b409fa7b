#!/bin/bash
# Round 6, call 20: the suite, 1000 fuzz seeds and the bench lines (driver's command, configs 5 and 2) on the tree with the host-side pools
# (own_bvh.h's top, tree_rotate.h's passes, occluder_first, the weights), plus where bench.py's scene generator spends scene_s.
O=gpurun_out/r06_call20; mkdir -p $O
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
P="import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); c=d.get('cold_job') or {}; pf=d.get('per_frame') or {}; mc=pf.get('moving_camera') or {}
print(d['value'], d['ms_per_step'], 'adapted in', d['adaptation'].get('seconds_to_adapted'), 'per frame', pf.get('ms_per_frame'), 'moving', mc.get('ms_per_frame'), mc.get('with_over_without'), 'parity', (d.get('parity') or {}).get('bit_identical'), 'roofline', (d.get('roofline') or {}).get('frac'), (d.get('roofline') or {}).get('stale'), 'cold', {k: v for k, v in c.items() if k not in ('what', 'trees', 'full_batch', 'rays')}, 'setup', d['config'].get('setup_s'), d['config'].get('scene_s'), d['config'].get('setup_breakdown'))"
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; el suite: $(grep -E "passed|failed" $O/pytest_gpu.log | tail -1)
RT_FUZZ_SEEDS=1000 timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -n 32 -p no:cacheprovider > $O/fuzz_1000_seeds.log 2>&1; el fuzz: $(tail -1 $O/fuzz_1000_seeds.log)
( time timeout 900 python bench.py > $O/bench_driver_command.json 2>> $O/bench.err ) 2>&1 | grep real; el bench: $(python -c "$P" $O/bench_driver_command.json 2>&1 | tail -1)
for cfg in 5 2; do
  timeout 900 python bench.py --config $cfg --no-cpu-baseline --surface-area-fold-steps 0 > $O/bench_cfg$cfg.json 2>> $O/bench.err; el cfg $cfg: $(python -c "$P" $O/bench_cfg$cfg.json 2>&1 | tail -1)
done
python - > $O/scene_profile.log 2>&1 <<'PY'
import cProfile, pstats, time, io
from raytracing_amd import scenes as S
t = time.time(); S.city_block(2_800_000); print("city_block: %.3f s" % (time.time() - t))
pr = cProfile.Profile(); pr.enable(); S.city_block(2_800_000); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(14); print(s.getvalue()[:3000])
PY
head -30 $O/scene_profile.log
grep -v amdgpu.ids $O/bench.err | tail -5
