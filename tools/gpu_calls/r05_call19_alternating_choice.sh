#!/bin/bash
# Round 5, call 19: the measured choice with the two ways ALTERNATING over frames 4 - 19 (call 18: timed in two blocks, a cold process chose wrongly).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_call19
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 600 python -m pytest tests/test_gpu_frame_kernel.py -q -m gpu -p no:cacheprovider > $O/pytest_frame_kernel.log 2>&1; el frame kernel tests: $(tail -1 $O/pytest_frame_kernel.log); grep -E "^E " $O/pytest_frame_kernel.log | head -10
python - <<PY > $O/make_cache.log 2>&1
import argparse, bench
from raytracing_amd import host, scenes as S
for cfg in (4, 2, 3):
    c = bench.CONFIGS[cfg]
    args = argparse.Namespace(config=cfg, scene=None, blob_tris=871_200, ball_tris=20_000, width=c["width"], height=c["height"], bounces=c["bounces"])
    raw = bench.build_scene(args, host, S, finish=False); raw.save_cache("/tmp/cfg%d.rtscene" % cfg); raw.close()
PY
for rep in 1 2; do
timeout 300 raytracing_amd/rt_render -w 1920 -h 1080 --scene /tmp/cfg4.rtscene --bounces 8 --frames 192 2>&1 | tee -a $O/rt_render_frames.log | tail -1
timeout 300 raytracing_amd/rt_render -w 1280 -h 720 --scene /tmp/cfg2.rtscene --bounces 8 --frames 192 2>&1 | tee -a $O/rt_render_frames.log | tail -1
done
timeout 300 raytracing_amd/rt_render -w 1920 -h 1080 --scene /tmp/cfg3.rtscene --bounces 3 --frames 192 2>&1 | tee -a $O/rt_render_frames.log | tail -1
el rt_render
for cfg in 4 2; do timeout 300 python bench.py --config $cfg --per-frame-only --per-frame-frames 96 --moving-camera-frames 0 > $O/pf_cfg${cfg}.json 2>> $O/bench.err; python -c "
import json; d=json.loads(open('$O/pf_cfg${cfg}.json').read().strip().splitlines()[-1]); p=d['per_frame']; print('cfg $cfg default (measured choice):', p['ms_per_frame'], 'ms per frame,', p['frames_through_k_frame'], 'of', p['frames'], 'frames through k_frame')"; done
el all done
