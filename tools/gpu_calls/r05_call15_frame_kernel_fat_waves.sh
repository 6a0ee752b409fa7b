#!/bin/bash
# Round 5, call 15: k_frame with MORE chunks per wave than the resident grid gives (fewer, fatter waves: better lane refill, less latency hiding).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_call15
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
for cfg in 4 2; do
  for fk in 1 8 10 12 16 24; do
    timeout 300 python bench.py --config $cfg --per-frame-only --per-frame-frames 96 --moving-camera-frames 0 --frame-kernel $fk > $O/pf_cfg${cfg}_fk$fk.json 2>> $O/bench.err; el $(pf pf_cfg${cfg}_fk$fk)
  done
done
