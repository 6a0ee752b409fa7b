#!/bin/bash
# Round 6, call 4: one fold adaptation per process group (rt_scene_export_folds / rt_scene_import_folds: two contexts, bench.py --gpus 2 and --gpus 8 on the one GPU),
# the fuzz campaign with RT_OPT_SAMPLES_AHEAD among its variants, the bench line with its new objects (cold_job, setup_breakdown, per_frame.samples_ahead,
# roofline.hbm_all_kernels), the compact log against the full one, rt_render --frames with and without samples ahead.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_call04
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 900 python -m pytest tests/test_gpu_device_fold.py tests/test_gpu_bench_scene.py tests/test_gpu_samples_ahead.py -q -m gpu -p no:cacheprovider > $O/pytest_new.log 2>&1; el new tests: $(grep -aE "passed|failed|rror" $O/pytest_new.log | tail -1)
grep -aE "^E  |^FAILED" $O/pytest_new.log | head -30
RT_FUZZ_SEEDS=700 timeout 600 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -n 32 -p no:cacheprovider > $O/fuzz_700_seeds.log 2>&1; el fuzz: $(tail -1 $O/fuzz_700_seeds.log)
grep -aE "^E  |^FAILED" $O/fuzz_700_seeds.log | head -10
( time python bench.py ) > $O/bench_driver_command.json 2> $O/bench.err; el bench: $(python -c "
import json; d=json.loads(open('$O/bench_driver_command.json').read().strip().splitlines()[-1]); p=d['per_frame']; a=p.get('samples_ahead') or {}; r=d['roofline']
print(d['value'], d['ms_per_step'], 'per frame', p['ms_per_frame'], p['mrays_per_s'], a.get('ms_per_call_median'), a.get('ms_per_call_p99'), a.get('bit_identical_to_rt_integrate_of_the_same_samples'), 'one sample per call', p['frame_kernel'].get('one_sample_per_call'), 'moving', p['moving_camera']['ms_per_frame'], p['moving_camera']['with_over_without'], 'parity', d['parity']['bit_identical'], 'roofline', r.get('frac'), r.get('stale'), 'cold', d['cold_job'], 'setup', d['config'].get('setup_breakdown'), 'adapt', d['adaptation'])" 2>&1 | tail -1)
grep real $O/bench.err
ARGS="--steps 4 --warmup 1 --no-cpu-baseline --per-frame-frames 0 --surface-area-fold-steps 0 --cold-job-spp 0"
for v in 0 1; do
  timeout 300 python bench.py $ARGS --compact-log $v > $O/compact_log_$v.json 2>> $O/bench.err
  el compact log $v: $(python -c "
import json; d=json.loads(open('$O/compact_log_$v.json').read().strip().splitlines()[-1]); print(d['value'], d['config']['path_state_GB'], d['config']['samples_in_flight'], d['roofline']['live_isolated']['kernel_ms_per_spp'])" 2>&1 | tail -1)
done
python - <<PY > $O/make_cache.log 2>&1
import argparse, bench
from raytracing_amd import host, scenes as S
c = bench.CONFIGS[4]
args = argparse.Namespace(config=4, scene=None, blob_tris=871_200, ball_tris=20_000, width=c["width"], height=c["height"], bounces=c["bounces"])
raw = bench.build_scene(args, host, S, finish=False); raw.save_cache("/tmp/cfg4.rtscene"); raw.close()
PY
for a in 1 0; do timeout 300 raytracing_amd/rt_render -w 1920 -h 1080 --scene /tmp/cfg4.rtscene --bounces 8 --frames 192 --samples_ahead $a 2>&1 | tee $O/rt_render_frames_cfg4_ahead$a.log | tail -1; done
el all done
