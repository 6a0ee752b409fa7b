#!/bin/bash
# Round 5, call 11: k_shade compiled for 7 / 8 waves per SIMD (72 / 64 VGPRs with 5 / 14 dwords spilled, instead of 78 at 6) and 256-thread blocks
# (so that 7 per SIMD is a whole number of blocks per CU): does more residency hide its gathers?  Same box, the shipped library between the variants.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_call11
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
line() { python - <<PY
import json
try:
    d = json.loads(open("$O/$1.json").read().strip().splitlines()[-1])
    k = (d["roofline"].get("live_isolated") or d["roofline"]["live"])["kernel_ms_per_spp"]
    print("$1: %.1f Mrays/s, alone %s" % (d["value"], k))
except Exception as e:
    print("$1: FAILED", e)
PY
}
Q="--steps 4 --no-cpu-baseline --per-frame-frames 0 --surface-area-fold-steps 0 --moving-camera-frames 0"
cp raytracing_amd/librt_hip.so $O/librt_hip_current.so
for v in current r05_shade8 r05_shade7_b256 r05_shade8_b256 r05_shade6_b256 current r05_shade8; do
  if [ $v = current ]; then cp $O/librt_hip_current.so raytracing_amd/librt_hip.so; else cp raytracing_amd/variants/$v/librt_hip.so raytracing_amd/librt_hip.so; fi
  n=$(ls $O | grep -c "^v_${v}_")
  timeout 300 python bench.py $Q > $O/v_${v}_$n.json 2>> $O/bench.err; el $(line v_${v}_$n)
done
cp $O/librt_hip_current.so raytracing_amd/librt_hip.so
rm -f $O/librt_hip_current.so
