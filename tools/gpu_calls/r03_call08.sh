#!/bin/bash
# round 3, GPU call 8: the whole GPU suite (no -x) after the fixes of call 7, headline with the fill-bounded automatic batch
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_call08
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=6 2>&1 | grep -aE "passed|failed|rror|FAILED|assert|s call|s setup" | tail -24 > $O/pytest_gpu.log; el suite: $(grep -aE "passed|failed" $O/pytest_gpu.log | tail -1); cat $O/pytest_gpu.log
b() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1]); k=(d["roofline"].get("live_isolated") or d["roofline"]["live"])["kernel_ms_per_spp"]
    print("$name: %.1f Mrays/s %.4f ms/spp in flight %d (%.1f GB) | alone: closest %.4f shadow %.4f shade %.4f | per-frame %s" % (d["value"], d["ms_per_spp"], d["config"]["samples_in_flight"], d["config"]["path_state_GB"], k["trace_closest"], k["trace_shadow"], k["shade"], (d.get("per_frame") or {}).get("mrays_per_s")))
except Exception as e:
    print("$name: FAILED", e); print(open("$O/bench_$name.err").read()[-800:])
PY
}
b cfg4; el cfg4
b cfg2 --config 2 --steps 4 --per-frame-frames 0; el cfg2
b cfg3 --config 3 --steps 4 --per-frame-frames 0; el cfg3
el all done
