#!/bin/bash
# Round 5, call 2: the asynchronous probe / pointer-exchange adoption and RT_OPT_STAGE_PIPES on the device -- their tests first, then the per-frame
# pattern with the frame's one sample per pixel on 1 .. 4 pipes, a turning camera, the driver's bench command on the new defaults, and the
# N-way tile timing behind `scaling_estimate`.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_call02
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline_parity.py -x -q -m gpu -k "adaptive_fold or stage or present" -p no:cacheprovider > $O/pytest_changed_paths.log 2>&1; el changed paths: $(tail -1 $O/pytest_changed_paths.log)
pf() { python - <<PY
import json
try:
    d = json.loads(open("$O/$1.json").read().strip().splitlines()[-1])
    p = d["per_frame"]; m = p.get("moving_camera") or {}
    print("$1: %.1f Mrays/s, %.3f ms per frame | moving %s ms, not moved %s ms, adoptions %s" % (p["mrays_per_s"], p["ms_per_frame"], m.get("ms_per_frame"), m.get("ms_per_frame_camera_set_not_moved"), m.get("adaptations_adopted_meanwhile")))
except Exception as e:
    print("$1: FAILED", e)
PY
}
for sp in 1 2 3 4; do
  mc=0; [ $sp = 1 ] && mc=240; [ $sp = 4 ] && mc=240
  timeout 300 python bench.py --per-frame-only --per-frame-frames 96 --stage-pipes $sp --moving-camera-frames $mc > $O/pf_cfg4_pipes$sp.json 2>> $O/bench.err; el $(pf pf_cfg4_pipes$sp)
done
for cfg in 2 3 5; do for sp in 1 4; do
  timeout 300 python bench.py --config $cfg --per-frame-only --per-frame-frames 96 --stage-pipes $sp --moving-camera-frames 0 > $O/pf_cfg${cfg}_pipes$sp.json 2>> $O/bench.err; el $(pf pf_cfg${cfg}_pipes$sp)
done; done
timeout 400 python bench.py > $O/bench_default.json 2>> $O/bench.err
python - <<PY
import json
d = json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("bench default: %.1f Mrays/s, per frame %s, parity %s, self %s, adaptation %s, SA fold %s" % (d["value"], d["per_frame"], {k: d["parity"].get(k) for k in ("bit_identical", "rel_l2_vs_libm_build", "reference_self_rel_l2", "median_pixel_rel_err_vs_libm_build", "reference_self_median_pixel_rel_err")}, None, d.get("adaptation"), d.get("surface_area_fold")))
PY
el bench default
timeout 400 python tools/tile_efficiency.py --json $O/tile_efficiency.json > $O/tile_efficiency.log 2>> $O/bench.err; cat $O/tile_efficiency.log; el tile efficiency
