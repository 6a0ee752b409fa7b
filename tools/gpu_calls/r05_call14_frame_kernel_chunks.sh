#!/bin/bash
# Round 5, call 14: k_frame (5 waves per SIMD now) with all its blocks resident (RT_OPT_FRAME_KERNEL = 1) against 4 / 2 chunks per wave (more blocks than
# are resident: the hardware's block scheduler balances the frame's chunks), per-frame leg on configs 4 / 2 / 3 / 5; the tests at 2 chunks per wave.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_call14
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
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
for cfg in 4 2 3 5; do
  for fk in 1 4 2 3; do
    timeout 300 python bench.py --config $cfg --per-frame-only --per-frame-frames 96 --moving-camera-frames 0 --frame-kernel $fk > $O/pf_cfg${cfg}_fk$fk.json 2>> $O/bench.err; el $(pf pf_cfg${cfg}_fk$fk)
  done
done
RT_TEST_FRAME_KERNEL_VALUE=2 timeout 900 python -m pytest tests/test_gpu_frame_kernel.py -x -q -m gpu -p no:cacheprovider > $O/pytest_frame_kernel_k2.log 2>&1; el frame kernel tests at 2 chunks per wave: $(tail -1 $O/pytest_frame_kernel_k2.log); grep -E "^E " $O/pytest_frame_kernel_k2.log | head -10
