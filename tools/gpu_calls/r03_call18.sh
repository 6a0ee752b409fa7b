#!/bin/bash
# round 3, GPU call 18: the compact radiance log (six inline entries + overflow pool, fallback to the full layout): parity and cost
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_call18
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 300 python -m pytest tests/test_gpu_headline_parity.py -q -m gpu -x -p no:cacheprovider -k "compact_radiance" > $O/pytest_compact.log 2>&1; el compact test: $(tail -1 $O/pytest_compact.log); grep -aE "Error|assert|error" $O/pytest_compact.log | head -10
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest_gpu_full.log 2>&1; el suite: $(grep -aE "passed|failed" $O/pytest_gpu_full.log | tail -1); grep -aE "^FAILED|^ERROR" $O/pytest_gpu_full.log | head
( RT_FUZZ_SEEDS=1500 timeout 400 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -n 32 -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror|Timeout" | tail -3 ) > $O/fuzz_1500_seeds.log 2>&1; el fuzz: $(tail -1 $O/fuzz_1500_seeds.log)
b() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --per-frame-frames 0 "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1]); k=(d["roofline"].get("live_isolated") or d["roofline"]["live"])["kernel_ms_per_spp"]
    print("$name: %.1f Mrays/s %.4f ms/spp in flight %d (%.1f GiB) | alone: closest %.4f shadow %.4f shade %.4f" % (d["value"], d["ms_per_spp"], d["config"]["samples_in_flight"], d["config"]["path_state_GB"], k["trace_closest"], k["trace_shadow"], k["shade"]))
except Exception as e:
    print("$name: FAILED", e); print(open("$O/bench_$name.err").read()[-800:])
PY
}
b cfg4_compact --steps 4; el
b cfg4_full --steps 4 --compact-log 0; el
b cfg4_compact_again --steps 4; el
b cfg5_compact --config 5 --steps 3; el
b cfg5_full --config 5 --steps 3 --compact-log 0; el
b cfg2_compact --config 2 --steps 4; el
b cfg4_32gb --steps 4 --path-state-gb 32; el
el all done
