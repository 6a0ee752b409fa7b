#!/bin/bash
# (Needs the experiment's switch -- RT_CTX_OPT_DEVICE_FOLD = 2 = the asynchronous adaptation's folds on the device -- which was removed after this call: worse both ways.)
# Round 6, call 24: the asynchronous adaptation's folds on the device again (RT_CTX_OPT_DEVICE_FOLD = 2), now that the worker's other stages are short: when does the adapted
# fold land, and what do the frames beside it pay (the moving-camera leg)?  Call 4 measured + 23 % for this; policy since: host threads when nothing waits.
O=gpurun_out/r06_call24; mkdir -p $O
for df in 1 2 1 2; do
  timeout 300 python tools/async_adaptation_time.py --config 4 --device-fold $df 2>&1 | tail -1 | cut -c1-420
  timeout 300 python bench.py --per-frame-only --per-frame-frames 48 --moving-camera-frames 720 --device-fold $df 2>> $O/err.log | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); m=d['per_frame']['moving_camera']; print('device fold $df: per frame', d['per_frame']['ms_per_frame'], 'moving', {k: m[k] for k in m if k in ('ms_per_frame','with_over_without','adaptations_adopted','ms_per_frame_without_re_adaptation')})"
done
timeout 300 python tools/async_adaptation_time.py --config 5 --device-fold 2 2>&1 | tail -1 | cut -c1-420
grep -v amdgpu.ids $O/err.log | tail -3
