#!/bin/bash
# Round 5, call 4: same-box A/B of the library before the refill quorum existed (commit 31b6ecd, built as raytracing_amd/variants/r05_pre_quorum)
# against the current one (quorum 16 by default) -- did the extra scalar work in loop C cost the hot loop anything? -- then the probe for two
# frames of the frame-by-frame pattern in flight (two Render objects taking turns).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_call04
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
for round in 1 2; do
  cp raytracing_amd/variants/r05_pre_quorum/librt_hip.so raytracing_amd/librt_hip.so
  timeout 300 python bench.py $Q > $O/ab_pre_quorum_$round.json 2>> $O/bench.err; el $(line ab_pre_quorum_$round)
  cp $O/librt_hip_current.so raytracing_amd/librt_hip.so
  timeout 300 python bench.py $Q > $O/ab_current_$round.json 2>> $O/bench.err; el $(line ab_current_$round)
  timeout 300 python bench.py $Q --refill-quorum 1 > $O/ab_current_rq1_$round.json 2>> $O/bench.err; el $(line ab_current_rq1_$round)
done
rm -f $O/librt_hip_current.so
timeout 400 python tools/two_frames_in_flight_probe.py > $O/two_frames_probe.log 2>> $O/bench.err; tail -3 $O/two_frames_probe.log | cut -c1-400; el two frames probe
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "samples_in_flight_and_kernel_variants" -p no:cacheprovider -n 8 > $O/pytest_variants.log 2>&1; el variants with the quorum among them: $(tail -1 $O/pytest_variants.log)
