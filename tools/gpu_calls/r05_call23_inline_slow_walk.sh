#!/bin/bash
# Round 5, call 23: the instance of k_trace_w4 with loop D (small launches: the frame-by-frame pattern) walks the rays the wide walk does not take
# (non-finite 1 / dir) itself, so the k_trace2 follow-up launch goes (18 launches of a frame less); k_frame drops its slow list the same way.
# Same-box A/B against the library built from HEAD (raytracing_amd/variants/r05_head), alternating.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_call23
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
cp raytracing_amd/librt_hip.so $O/new.so
timeout 400 python -m pytest tests/test_gpu_frame_kernel.py tests/test_gpu_parity.py -x -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; el tests: $(tail -1 $O/pytest.log); grep -E "^E " $O/pytest.log | head -10
RT_FUZZ_SEEDS=300 timeout 300 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -p no:cacheprovider > $O/fuzz.log 2>&1; el fuzz: $(tail -1 $O/fuzz.log)
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
for lib in new head; do
  if [ $lib = head ]; then cp raytracing_amd/variants/r05_head/librt_hip.so raytracing_amd/librt_hip.so; else cp $O/new.so raytracing_amd/librt_hip.so; fi
  timeout 300 python bench.py --config 2 --per-frame-only --per-frame-frames 96 --moving-camera-frames 0 --frame-kernel 1 > $O/pf_cfg2_fk1_$lib.json 2>> $O/bench.err; el $(pf pf_cfg2_fk1_$lib)
  timeout 300 python bench.py --config 4 --per-frame-only --per-frame-frames 96 --moving-camera-frames 0 --frame-kernel 1 > $O/pf_cfg4_fk1_$lib.json 2>> $O/bench.err; el $(pf pf_cfg4_fk1_$lib)
done
cp $O/new.so raytracing_amd/librt_hip.so
rm -f $O/new.so
tail -3 $O/bench.err
