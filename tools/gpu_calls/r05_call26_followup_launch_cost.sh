#!/bin/bash
# Round 5, call 26: what do the 18 k_trace2 follow-up launches of a frame COST?  (Calls 23 / 24 tried to remove them and lost to register pressure; the ~2 % they
# were after was an estimate.)  A host-only variant that simply does not launch the follow-up behind the loop-D instance -- wrong for RT_SIGN_SLOW rays, so a
# measurement only, never a product: built from HEAD + one line (`if (tail) return;` before the launch in launch_trace_w4), same device code object.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_call26
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
cp raytracing_amd/librt_hip.so $O/head.so
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
for rep in 1 2 3; do
  for lib in skip head; do
    if [ $lib = skip ]; then cp raytracing_amd/variants/r05_skip_followup/librt_hip.so raytracing_amd/librt_hip.so; else cp $O/head.so raytracing_amd/librt_hip.so; fi
    timeout 300 python bench.py --config 4 --per-frame-only --per-frame-frames 96 --moving-camera-frames 0 --frame-kernel 0 > $O/pf_cfg4_fk0_${lib}_$rep.json 2>> $O/bench.err; el $(pf pf_cfg4_fk0_${lib}_$rep)
  done
done
for lib in skip head; do
  if [ $lib = skip ]; then cp raytracing_amd/variants/r05_skip_followup/librt_hip.so raytracing_amd/librt_hip.so; else cp $O/head.so raytracing_amd/librt_hip.so; fi
  timeout 300 python bench.py --config 2 --per-frame-only --per-frame-frames 96 --moving-camera-frames 0 --frame-kernel 0 > $O/pf_cfg2_fk0_$lib.json 2>> $O/bench.err; el $(pf pf_cfg2_fk0_$lib)
done
cp $O/head.so raytracing_amd/librt_hip.so; rm -f $O/head.so
