#!/bin/bash
# Round 5, call 13: k_frame with a wave's chunks taken from the WHOLE tile (default now) against its XCD's eighth (call 12), and compiled for 5 waves
# per SIMD (96 VGPRs, 27 dwords spilled) instead of 4 (126): tests on the default, then the per-frame leg on configs 4 / 2 / 3 / 5.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_call13
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 900 python -m pytest tests/test_gpu_frame_kernel.py -x -q -m gpu -p no:cacheprovider > $O/pytest_frame_kernel.log 2>&1; el frame kernel tests: $(tail -1 $O/pytest_frame_kernel.log); grep -E "^E " $O/pytest_frame_kernel.log | head -10
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
cp raytracing_amd/librt_hip.so $O/librt_hip_current.so
for cfg in 4 2 3 5; do
  timeout 300 python bench.py --config $cfg --per-frame-only --per-frame-frames 96 --moving-camera-frames 0 --frame-kernel 0 > $O/pf_cfg${cfg}_stage.json 2>> $O/bench.err; el $(pf pf_cfg${cfg}_stage)
  for v in current r05_frame_xcd r05_frame_w5; do
    if [ $v = current ]; then cp $O/librt_hip_current.so raytracing_amd/librt_hip.so; else cp raytracing_amd/variants/$v/librt_hip.so raytracing_amd/librt_hip.so; fi
    timeout 300 python bench.py --config $cfg --per-frame-only --per-frame-frames 96 --moving-camera-frames 0 --frame-kernel 1 > $O/pf_cfg${cfg}_$v.json 2>> $O/bench.err; el $(pf pf_cfg${cfg}_$v)
  done
  cp $O/librt_hip_current.so raytracing_amd/librt_hip.so
done
rm -f $O/librt_hip_current.so
