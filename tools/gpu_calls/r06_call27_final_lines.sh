#!/bin/bash
# Round 6, call 27: call 10's commands again on the final tree (after the host-side pools -- own_bvh.h, tree_rotate.h, build_wide_bvh --, the cold job first in the bench, the full-size tests):
# the driver's commands, every config's bench line with its CPU leg, rt_render (frames and a cold 256-spp job from C++).  The hot path's code object is call 5's
# (d16574bc...: its counters stay valid).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_call27
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 1200 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/pytest_gpu.log 2>&1; el suite: $(grep -aE "passed|failed|rror" $O/pytest_gpu.log | tail -1)
grep -aE "^E  |^FAILED" $O/pytest_gpu.log | head -10
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; el smoke: $(tail -1 $O/smoke.log)
RT_FUZZ_SEEDS=2000 timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -n 32 -p no:cacheprovider > $O/fuzz_2000_seeds.log 2>&1; el fuzz: $(tail -1 $O/fuzz_2000_seeds.log)
( time python bench.py ) > $O/bench_driver_command.json 2> $O/bench.err; el bench: $(python -c "
import json; d=json.loads(open('$O/bench_driver_command.json').read().strip().splitlines()[-1]); p=d['per_frame']; a=p.get('samples_ahead') or {}; r=d['roofline']; c=d['config']
print(d['value'], d['ms_per_step'], 'per frame', p['ms_per_frame'], p['mrays_per_s'], a.get('ms_per_call_median'), a.get('ms_per_call_p99'), a.get('bit_identical_to_rt_integrate_of_the_same_samples'), 'one per call', p['frame_kernel']['one_sample_per_call']['ms_per_frame'], 'moving', p['moving_camera']['ms_per_frame'], p['moving_camera']['with_over_without'], 'parity', d['parity']['bit_identical'], 'roofline', r.get('frac'), r.get('stale'), 'cold', {k: v for k, v in (d['cold_job'] or {}).items() if k not in ('what', 'trees')}, 'setup', c.get('setup_s'), c.get('setup_breakdown'), c.get('path_state_alloc_s'), 'adapt', d['adaptation'].get('seconds_to_adapted'), 'sa fold', d['surface_area_fold'].get('value'), 'cpu', d['cpu_baseline']['value'])" 2>&1 | tail -1)
grep real $O/bench.err
for cfg in 2 3 1 5; do
  extra=""; [ $cfg = 1 ] && extra="--steps 64 --warmup 4 --per-frame-frames 192"; [ $cfg = 5 ] && extra="--cpu-seconds 5"
  timeout 700 python bench.py --config $cfg $extra > $O/bench_cfg$cfg.json 2>> $O/bench.err; el cfg $cfg: $(python -c "
import json; d=json.loads(open('$O/bench_cfg$cfg.json').read().strip().splitlines()[-1]); p=d['parity']; f=d['per_frame']; a=f.get('samples_ahead') or {}; r=d['roofline']
print(d['value'], 'per frame', f['ms_per_frame'], a.get('bit_identical_to_rt_integrate_of_the_same_samples'), 'one per call', f['frame_kernel'].get('one_sample_per_call', {}).get('ms_per_frame'), f['frame_kernel'].get('default_went'), 'parity', p['bit_identical'], p.get('rel_l2_vs_libm_build'), 'roofline', r.get('frac'), r.get('stale'), 'sa fold', (d['surface_area_fold'] or {}).get('value'), 'cold', {k: v for k, v in (d['cold_job'] or {}).items() if k not in ('what', 'trees')})" 2>&1 | tail -1)
done
python - <<PY > $O/make_cache.log 2>&1
import argparse, bench
from raytracing_amd import host, scenes as S
for cfg in (4, 2):
    c = bench.CONFIGS[cfg]
    args = argparse.Namespace(config=cfg, scene=None, blob_tris=871_200, ball_tris=20_000, width=c["width"], height=c["height"], bounces=c["bounces"])
    raw = bench.build_scene(args, host, S, finish=False); raw.save_cache("/tmp/cfg%d.rtscene" % cfg); raw.close()
PY
for a in 1 0; do timeout 300 raytracing_amd/rt_render -w 1920 -h 1080 --scene /tmp/cfg4.rtscene --bounces 8 --frames 192 --samples_ahead $a 2>&1 | tee $O/rt_render_frames_cfg4_ahead$a.log | tail -1; done
timeout 300 raytracing_amd/rt_render -w 1280 -h 720 --scene /tmp/cfg2.rtscene --bounces 8 --frames 192 2>&1 | tee $O/rt_render_frames_cfg2.log | tail -1
( time timeout 300 raytracing_amd/rt_render -w 1920 -h 1080 --scene /tmp/cfg4.rtscene --bounces 8 --spp 256 ) 2>&1 | tee $O/rt_render_cold_256spp_cfg4.log | tail -5
el all done
