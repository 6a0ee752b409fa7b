#!/bin/bash
# PREPARED in round 4 (its GPU minutes were spent), to be the first call of round 5: the fold adapted to the frame's rays against the upload's
# surface-area fold on EVERY config (round 4 measured the headline only: 6340 -> 6707 Mrays/s on one box), each line with its parity leg, then
# the per-frame leg alone with both folds, then a moved camera (does the re-adaptation pay within a flight?).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_call01
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
line() { python - <<PY
import json
try:
    d = json.loads(open("$O/$1.json").read().strip().splitlines()[-1])
    p, pf = d.get("parity") or {}, d.get("per_frame") or {}
    k = (d["roofline"].get("live_isolated") or d["roofline"]["live"])["kernel_ms_per_spp"]
    print("$1: %.1f Mrays/s, per frame %s, bit_identical=%s, alone %s | %s" % (d["value"], pf.get("mrays_per_s"), p.get("bit_identical"), k, d["config"]["trees"][-1][:160]))
except Exception as e:
    print("$1: FAILED", e)
PY
}
RT_TEST_ADAPTIVE_BIT4=1 timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k adaptive_fold -p no:cacheprovider > $O/pytest_adaptive_fold_all_modes.log 2>&1; el adaptive-fold tests, bit 4 included: $(tail -1 $O/pytest_adaptive_fold_all_modes.log)
for cfg in 4 2 3 1 5; do
  for fold in 3 11 27 0; do   # 3: the adapted fold (default), 11: + the shadow rays' tree rotated first (bit 3), 27: + slots likeliest occluder first (bit 4) -- both untimed so far, 0: the upload's fold
    extra=""; [ $cfg = 1 ] && extra="--steps 64 --warmup 4"; [ $cfg = 5 ] && extra="--cpu-seconds 5"
    python bench.py --config $cfg --adaptive-fold $fold $extra > $O/bench_cfg${cfg}_fold${fold}.json 2>> $O/bench.err; el $(line bench_cfg${cfg}_fold${fold})
  done
done
el all done
