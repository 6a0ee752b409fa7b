#!/bin/bash
# Round 5, call 12: k_frame (RT_OPT_FRAME_KERNEL) on the device for the first time: its tests, then the per-frame leg with and without it.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_call12
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 900 python -m pytest tests/test_gpu_frame_kernel.py -x -q -m gpu -p no:cacheprovider > $O/pytest_frame_kernel.log 2>&1; el frame kernel tests: $(tail -1 $O/pytest_frame_kernel.log); grep -E "^E " $O/pytest_frame_kernel.log | head -20
pf() { python - <<PY
import json
try:
    d = json.loads(open("$O/$1.json").read().strip().splitlines()[-1])
    p = d["per_frame"]
    print("$1: %.1f Mrays/s, %.3f ms per frame" % (p["mrays_per_s"], p["ms_per_frame"]))
except Exception as e:
    print("$1: FAILED", e)
PY
}
for fk in 0 1; do
  timeout 300 python bench.py --per-frame-only --per-frame-frames 96 --moving-camera-frames 0 --frame-kernel $fk > $O/pf_cfg4_fk$fk.json 2>> $O/bench.err; el $(pf pf_cfg4_fk$fk)
done
for cfg in 2 3; do for fk in 0 1; do
  timeout 300 python bench.py --config $cfg --per-frame-only --per-frame-frames 96 --moving-camera-frames 0 --frame-kernel $fk > $O/pf_cfg${cfg}_fk$fk.json 2>> $O/bench.err; el $(pf pf_cfg${cfg}_fk$fk)
done; done
tail -3 $O/bench.err | cut -c1-300
