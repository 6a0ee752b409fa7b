#!/bin/bash
# Round 5, call 24: call 23's inline slow walk made the instance of k_trace_w4 with loop D spill into its hot loops (frame-by-frame pattern 3.32 -> 3.73 ms
# per 1080p frame).  Here the walk is a noinline CALL (v1_trace_ray_call).  Same-box A/B against the library built from HEAD, alternating.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_call24
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
cp raytracing_amd/librt_hip.so $O/new.so
timeout 300 python -m pytest tests/test_gpu_frame_kernel.py -x -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; el tests: $(grep -E "passed|failed|error" $O/pytest.log | tail -1)
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
for rep in 1 2; do
  for lib in new head; do
    if [ $lib = head ]; then cp raytracing_amd/variants/r05_head/librt_hip.so raytracing_amd/librt_hip.so; else cp $O/new.so raytracing_amd/librt_hip.so; fi
    timeout 300 python bench.py --config 4 --per-frame-only --per-frame-frames 96 --moving-camera-frames 0 --frame-kernel 0 > $O/pf_cfg4_fk0_${lib}_$rep.json 2>> $O/bench.err; el $(pf pf_cfg4_fk0_${lib}_$rep)
  done
done
cp $O/new.so raytracing_amd/librt_hip.so
rm -f $O/new.so
