#!/bin/bash
# Round 5, call 18: HIPPathTraceIntegrator asks for the measured choice (RT_OPT_FRAME_KERNEL = 255) by default: the whole GPU suite on that, smoke, the
# driver's bench command (per_frame = the default; both ways forced beside it), configs 2 / 3 / 5 / 1 likewise, rt_render --frames.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_call18
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/pytest_gpu.log 2>&1; el suite: $(grep -aE "passed|failed|rror" $O/pytest_gpu.log | tail -1); grep -E "^E |^FAILED" $O/pytest_gpu.log | head -10
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; el smoke: $(tail -1 $O/smoke.log)
( time python bench.py ) > $O/bench_default.json 2> $O/bench.err; python - <<PY
import json
d = json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
p = d["per_frame"]
print("bench default: %.1f Mrays/s; per frame %.3f ms (%d of %d frames through k_frame); %s" % (d["value"], p["ms_per_frame"], p["frames_through_k_frame"], p["frames"], json.dumps({k: v for k, v in p["frame_kernel"].items() if k != "what"})))
print("   moving camera", p["moving_camera"]["ms_per_frame"], p["moving_camera"]["ms_per_frame_without_re_adaptation"], "parity", d["parity"]["bit_identical"], "stale", d["roofline"]["stale"])
PY
grep real $O/bench.err; el bench default
for cfg in 2 3 5 1; do
  extra=""; [ $cfg = 1 ] && extra="--steps 64 --warmup 4"
  timeout 400 python bench.py --config $cfg --no-cpu-baseline --surface-area-fold-steps 0 --moving-camera-frames 0 $extra > $O/bench_cfg$cfg.json 2>> $O/bench.err; python - <<PY
import json
d = json.loads(open("$O/bench_cfg$cfg.json").read().strip().splitlines()[-1])
p = d["per_frame"]; k = p["frame_kernel"]
print("cfg $cfg: %.1f Mrays/s; per frame %.3f ms (default went %s); stage kernels %s ms, k_frame %s ms; bit-identical %s %s" % (d["value"], p["ms_per_frame"], k["default_went"], k["stage_kernels"]["ms_per_frame"], k["k_frame"]["ms_per_frame"], k["stage_kernels"]["bit_identical_to_the_default_leg"], k["k_frame"]["bit_identical_to_the_default_leg"]))
PY
  el cfg $cfg
done
python - <<PY > $O/make_cache.log 2>&1
import argparse, bench
from raytracing_amd import host, scenes as S
for cfg in (4, 2):
    c = bench.CONFIGS[cfg]
    args = argparse.Namespace(config=cfg, scene=None, blob_tris=871_200, ball_tris=20_000, width=c["width"], height=c["height"], bounces=c["bounces"])
    raw = bench.build_scene(args, host, S, finish=False); raw.save_cache("/tmp/cfg%d.rtscene" % cfg); raw.close()
PY
timeout 300 raytracing_amd/rt_render -w 1920 -h 1080 --scene /tmp/cfg4.rtscene --bounces 8 --frames 192 2>&1 | tee $O/rt_render_frames_cfg4.log | tail -1
timeout 300 raytracing_amd/rt_render -w 1280 -h 720 --scene /tmp/cfg2.rtscene --bounces 8 --frames 192 2>&1 | tee $O/rt_render_frames_cfg2.log | tail -1
el all done
