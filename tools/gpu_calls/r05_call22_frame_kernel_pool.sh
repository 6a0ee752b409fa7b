#!/bin/bash
# Round 5, call 22: k_frame with a POOL -- the last rows of the chunk space are not owned by a wave but taken, a few chunks at a time, by the waves that
# have finished their own share.  RT_OPT_FRAME_KERNEL 16 + k: k pool rows (2 chunks per grab); 32 + k: 1 per grab; 48 + k: 4 per grab; 1: the default.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_call22
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 600 python -m pytest tests/test_gpu_frame_kernel.py -x -q -m gpu -p no:cacheprovider > $O/pytest_frame_kernel.log 2>&1; el frame kernel tests: $(tail -1 $O/pytest_frame_kernel.log); grep -E "^E " $O/pytest_frame_kernel.log | head -10
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
for cfg in 4 2 3; do
  for fk in 16 17 18 19 20 34 50 1; do
    timeout 300 python bench.py --config $cfg --per-frame-only --per-frame-frames 96 --moving-camera-frames 0 --frame-kernel $fk > $O/pf_cfg${cfg}_fk$fk.json 2>> $O/bench.err; el $(pf pf_cfg${cfg}_fk$fk)
  done
done
python tools/frame_kernel_rows.py --config 4 --value 1 2>&1 | grep -v amdgpu | tail -6
