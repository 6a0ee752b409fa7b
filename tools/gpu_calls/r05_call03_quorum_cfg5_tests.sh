#!/bin/bash
# Round 5, call 3: the changed paths' tests again (call 2 stopped at the first: restoring the context's option after an upload cleared the live
# wait bit -- now a separate option), the refill quorum of k_trace_w4 (a launch parameter: same code object for every value), config 5's
# candidates (16-entry LDS stack; compact log with more samples in flight), the moving-camera leg against the same path without re-adaptation.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_call03
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline_parity.py -q -m gpu -k "adaptive_fold or stage or present" -p no:cacheprovider > $O/pytest_changed_paths.log 2>&1; el changed paths: $(tail -1 $O/pytest_changed_paths.log)
line() { python - <<PY
import json
try:
    d = json.loads(open("$O/$1.json").read().strip().splitlines()[-1])
    k = (d["roofline"].get("live_isolated") or d["roofline"]["live"])["kernel_ms_per_spp"]
    print("$1: %.1f Mrays/s, alone %s, in flight %s, %s GB, spills %s" % (d["value"], k, d["config"]["samples_in_flight"], d["config"]["path_state_GB"], d["config"]["stack_spill_lane_steps"]))
except Exception as e:
    print("$1: FAILED", e)
PY
}
Q="--steps 4 --no-cpu-baseline --per-frame-frames 0 --surface-area-fold-steps 0 --moving-camera-frames 0"
for rq in 1 8 16 24 1; do
  timeout 300 python bench.py $Q --refill-quorum $rq > $O/q_cfg4_rq${rq}.json 2>> $O/bench.err; el $(line q_cfg4_rq${rq})
done
for rq in 1 16; do
  timeout 300 python bench.py $Q --refill-quorum $rq --trace-tune 0x0C28 > $O/q_cfg4_rq${rq}_40_12.json 2>> $O/bench.err; el $(line q_cfg4_rq${rq}_40_12)
done
timeout 300 python bench.py $Q --config 2 --refill-quorum 16 > $O/q_cfg2_rq16.json 2>> $O/bench.err; el $(line q_cfg2_rq16)
timeout 300 python bench.py $Q --config 2 --refill-quorum 1 > $O/q_cfg2_rq1.json 2>> $O/bench.err; el $(line q_cfg2_rq1)
for v in "" "--trace-variant 11" "--compact-log 1 --samples-in-flight 40" "--refill-quorum 16"; do
  n=$(echo "cfg5$v" | tr -d ' -' )
  timeout 400 python bench.py $Q --config 5 $v > $O/q_$n.json 2>> $O/bench.err; el $(line q_$n)
done
timeout 300 python bench.py --per-frame-only --per-frame-frames 96 > $O/pf_moving.json 2>> $O/bench.err; python -c "
import json; d = json.loads(open('$O/pf_moving.json').read().strip().splitlines()[-1]); print(json.dumps(d['per_frame'])[:900])"; el moving camera
