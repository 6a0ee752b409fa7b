#!/bin/bash
# Round 5, call 9: the frame-by-frame pattern from C++ (rt_render --frames: no Python, no PyTorch -- the system's HIP runtime, whose device-to-host
# copies go through the SDMA engines) against the same loop in bench.py's process (PyTorch's bundled HIP runtime, whose copies are shader blits that
# stall every store-heavy kernel meanwhile: tools/d2h_copy_probe.hip); which runtime is loaded, and what each does with the presented image.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_call09
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
TL=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))")
python - <<PY > $O/make_cache.log 2>&1
import argparse, bench
from raytracing_amd import host, scenes as S
for cfg in (4, 2):
    c = bench.CONFIGS[cfg]
    args = argparse.Namespace(config=cfg, scene=None, blob_tris=871_200, ball_tris=20_000, width=c["width"], height=c["height"], bounces=c["bounces"])
    raw = bench.build_scene(args, host, S, finish=False)
    raw.save_cache("/tmp/cfg%d.rtscene" % cfg)
    raw.close()
print("caches written")
PY
el $(tail -1 $O/make_cache.log)
echo "--- rt_render --frames (C++, system HIP runtime)"
timeout 300 raytracing_amd/rt_render -w 1920 -h 1080 --scene /tmp/cfg4.rtscene --bounces 8 --frames 192 2>&1 | tee $O/rt_render_frames_cfg4.log | tail -2
timeout 300 raytracing_amd/rt_render -w 1280 -h 720 --scene /tmp/cfg2.rtscene --bounces 8 --frames 192 2>&1 | tee $O/rt_render_frames_cfg2.log | tail -1
el rt_render
echo "--- the same binary on PyTorch's bundled HIP runtime (LD_LIBRARY_PATH=$TL)"
LD_LIBRARY_PATH=$TL timeout 300 raytracing_amd/rt_render -w 1920 -h 1080 --scene /tmp/cfg4.rtscene --bounces 8 --frames 192 2>&1 | tee $O/rt_render_frames_cfg4_torch_runtime.log | tail -1
el rt_render on the bundled runtime
echo "--- d2h probe on the bundled runtime"
LD_LIBRARY_PATH=$TL timeout 120 tools/bin/d2h_copy_probe 2>&1 | tee $O/probe_torch_runtime.log | grep -v "^$" | cut -c1-330
for knob in "GPU_FORCE_BLIT_COPY_SIZE=0" "HSA_ENABLE_SDMA=1" "GPU_BLIT_ENGINE_TYPE=1" "GPU_BLIT_ENGINE_TYPE=2"; do
  echo "--- bundled runtime with $knob"; env LD_LIBRARY_PATH=$TL $knob timeout 120 tools/bin/d2h_copy_probe 2>&1 | grep "rt_frame_present" | cut -c1-300
done
el probes
echo "--- bench.py per-frame leg: as it is, and with the system runtime preloaded"
timeout 300 python bench.py --per-frame-only --per-frame-frames 96 --moving-camera-frames 0 > $O/pf_python.json 2>> $O/bench.err; python -c "
import json; d=json.loads(open('$O/pf_python.json').read().strip().splitlines()[-1]); print('python + torch runtime:', d['per_frame']['ms_per_frame'], 'ms per frame', d['per_frame']['mrays_per_s'])"
LD_PRELOAD=/opt/rocm/lib/libamdhip64.so timeout 300 python bench.py --per-frame-only --per-frame-frames 96 --moving-camera-frames 0 > $O/pf_python_preload.json 2>> $O/bench_preload.err; python -c "
import json; d=json.loads(open('$O/pf_python_preload.json').read().strip().splitlines()[-1]); print('python, system runtime preloaded:', d['per_frame']['ms_per_frame'], 'ms per frame', d['per_frame']['mrays_per_s'])" 2>&1 | tail -1; tail -2 $O/bench_preload.err | cut -c1-300
el all done
