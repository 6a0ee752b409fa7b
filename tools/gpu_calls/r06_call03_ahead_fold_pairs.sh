#!/bin/bash
# Round 6, call 3: RT_OPT_SAMPLES_AHEAD after the camera fix (call 2: the integrator sets the same camera before every frame and that reset the quiet count:
# the mode never started through the hooks); RT_CTX_OPT_DEVICE_FOLD on the device for the first time (record-for-record against the host fold, the upload's
# stage times); RT_CTX_OPT_WIDE_LAYOUT = 1 (pairs) against 0 on configs 5 and 4; the leaf records' loads non-temporal (variant nt5) and the queue hint off (nt0).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_call03
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 600 python -m pytest tests/test_gpu_device_fold.py tests/test_gpu_samples_ahead.py tests/test_gpu_frame_kernel.py -q -m gpu -p no:cacheprovider > $O/pytest_new.log 2>&1; el new tests: $(grep -aE "passed|failed|rror" $O/pytest_new.log | tail -1)
grep -aE "^E  |^FAILED" $O/pytest_new.log | head -30
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest_gpu.log 2>&1; el suite: $(grep -aE "passed|failed|rror" $O/pytest_gpu.log | tail -1)
grep -aE "^FAILED" $O/pytest_gpu.log | head -12
pf() { # config, samples-ahead value, frames
  timeout 400 python bench.py --config $1 --per-frame-only --per-frame-frames $3 --moving-camera-frames 0 --samples-ahead $2 > $O/pf_cfg$1_ahead$2.json 2>> $O/bench.err
  el cfg $1 ahead $2: $(python -c "
import json; d=json.loads(open('$O/pf_cfg$1_ahead$2.json').read().strip().splitlines()[-1])['per_frame']; a=d.get('samples_ahead') or {}
print(d['ms_per_frame'], 'ms/frame', d['mrays_per_s'], 'Mrays/s', 'k_frame frames', d['frames_through_k_frame'], 'replayed', a.get('frames_replayed_from_a_batch'), 'median/p99/max', a.get('ms_per_call_median'), a.get('ms_per_call_p99'), a.get('ms_per_call_max'), 'same bits', a.get('bit_identical_to_rt_integrate_of_the_same_samples'))" 2>&1 | tail -1)
}
for a in 0 1 8 16 264 272 32; do pf 4 $a 192; done
for a in 1 2 260 8; do pf 5 $a 64; done
for a in 1 8 264; do pf 2 $a 192; pf 3 $a 192; done
pf 1 1 192
ARGS="--steps 4 --warmup 1 --no-cpu-baseline --per-frame-frames 0 --surface-area-fold-steps 0 --cold-job-spp 0"
run() { # name, config, extra args
  timeout 300 python bench.py --config $2 $ARGS $3 > $O/$1_cfg$2.json 2>> $O/bench.err
  el $1 cfg $2: $(python -c "
import json; d=json.loads(open('$O/$1_cfg$2.json').read().strip().splitlines()[-1]); k=d['roofline']['live_isolated']['kernel_ms_per_spp']; print(d['value'], k, d['config']['ranks'].get('setup_breakdown'), [t for t in d['config'].get('trees', []) if t.startswith('upload')])" 2>&1 | tail -1)
}
for rep in 1 2; do
  for cfg in 5 4; do
    run base_$rep $cfg ""
    run pairs_$rep $cfg "--wide-layout 1"
  done
done
run hostfold_1 4 "--device-fold 0"
run hostfold_1 5 "--device-fold 0"
cp raytracing_amd/librt_hip.so $O/base_librt_hip.so
for v in nt0 nt5; do
  cp raytracing_amd/variants/$v/librt_hip.so raytracing_amd/librt_hip.so
  for cfg in 5 4; do run ${v}_1 $cfg ""; run ${v}_2 $cfg ""; done
done
cp $O/base_librt_hip.so raytracing_amd/librt_hip.so; rm $O/base_librt_hip.so
el all done
