#!/bin/bash
# Round 5, call 21: k_frame with a wave's chunks shifted by 7 / 131 / 1021 columns per row of the chunk space (unshifted they sit under each other in
# one column of the image): per-frame leg with the kernel forced, configs 4 / 2 / 3; the frame-kernel tests on the best.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_call21
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
cp raytracing_amd/librt_hip.so $O/librt_hip_current.so
for cfg in 4 2 3; do
  for v in current r05_frame_shift7 r05_frame_shift131 r05_frame_shift1021; do
    if [ $v = current ]; then cp $O/librt_hip_current.so raytracing_amd/librt_hip.so; else cp raytracing_amd/variants/$v/librt_hip.so raytracing_amd/librt_hip.so; fi
    timeout 300 python bench.py --config $cfg --per-frame-only --per-frame-frames 96 --moving-camera-frames 0 --frame-kernel 1 > $O/pf_cfg${cfg}_$v.json 2>> $O/bench.err; el $(pf pf_cfg${cfg}_$v)
  done
done
cp raytracing_amd/variants/r05_frame_shift131/librt_hip.so raytracing_amd/librt_hip.so
timeout 600 python -m pytest tests/test_gpu_frame_kernel.py -q -m gpu -p no:cacheprovider > $O/pytest_frame_kernel_shift131.log 2>&1; el tests with shift 131: $(tail -1 $O/pytest_frame_kernel_shift131.log)
python tools/frame_kernel_rows.py --config 4 2>&1 | grep -v amdgpu | tail -6
cp $O/librt_hip_current.so raytracing_amd/librt_hip.so; rm -f $O/librt_hip_current.so
