for e in 0 1 0 1; do
  if [ $e = 1 ]; then export RT_EXPERIMENT_PLAIN_BESIDE=1; else unset RT_EXPERIMENT_PLAIN_BESIDE; fi
  timeout 300 python tools/async_adaptation_time.py --config 4 2>&1 | tail -1 | cut -c1-330
  timeout 300 python bench.py --per-frame-only --per-frame-frames 48 --moving-camera-frames 720 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); m=d['per_frame']['moving_camera']; print('beside=$e: moving', {k: m[k] for k in m if k in ('ms_per_frame','with_over_without','adaptations_adopted','ms_per_frame_without_re_adaptation')})"
done
