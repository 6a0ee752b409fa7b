#!/bin/bash
# Round 5, call 1 (prepared in round 4, trimmed to the round's GPU budget): the fold adapted to the frame's rays against the upload's
# surface-area fold on EVERY config, and bits 3 + 4 (rotated shadow tree, slots likeliest occluder first) on the device for the first
# time -- the parity leg (CPU reference render of the same frame) rides on the bit-4 line of each config, the others are timing only.
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
    print("$1: %.1f Mrays/s, per frame %s, bit_identical=%s, alone %s | %s" % (d["value"], pf.get("mrays_per_s"), p.get("bit_identical"), k, d["config"]["trees"][-1][:200]))
except Exception as e:
    print("$1: FAILED", e)
PY
}
RT_TEST_ADAPTIVE_BIT4=1 timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k adaptive_fold -p no:cacheprovider > $O/pytest_adaptive_fold_all_modes.log 2>&1; el adaptive-fold tests, bit 4 included: $(tail -1 $O/pytest_adaptive_fold_all_modes.log)
for cfg in 4 2 3 1 5; do
  for fold in 27 3 11 0; do   # 27: adapted fold + rotated shadow tree + occluder-first slots, 3: the adapted fold (r04 default), 11: bit 3 only, 0: the upload's fold
    [ $fold = 11 ] && [ $cfg != 4 ] && continue
    extra="--no-cpu-baseline"; [ $fold = 27 ] && extra=""
    [ $cfg = 1 ] && extra="$extra --steps 64 --warmup 4"; [ $cfg = 5 ] && [ $fold = 27 ] && extra="--cpu-seconds 5"
    timeout 400 python bench.py --config $cfg --adaptive-fold $fold $extra > $O/bench_cfg${cfg}_fold${fold}.json 2>> $O/bench.err; el $(line bench_cfg${cfg}_fold${fold})
  done
done
el all done
