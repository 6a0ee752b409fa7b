#!/bin/bash
# Round 5, call 5: the same A/B as call 4 after the quorum test was rewritten in mask form (call 4: written with a conditional it cost the node loop 3 %).
# against the current one (quorum 16 by default) -- did the extra scalar work in loop C cost the hot loop anything? -- then the probe for two
# frames of the frame-by-frame pattern in flight (two Render objects taking turns).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_call05
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
