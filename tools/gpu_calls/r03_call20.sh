#!/bin/bash
# round 3, GPU call 20: the suite and a fuzz campaign with the compact log among the variants (opt-in now); headline unchanged?
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_call20
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest_gpu_full.log 2>&1; el suite: $(grep -aE "passed|failed" $O/pytest_gpu_full.log | tail -1); grep -aE "^FAILED|^ERROR|^E  " $O/pytest_gpu_full.log | head
( RT_FUZZ_SEEDS=4000 timeout 600 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -n 32 -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror|Timeout" | tail -3 ) > $O/fuzz_4000_seeds.log 2>&1; el fuzz: $(tail -1 $O/fuzz_4000_seeds.log)
b() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1]); k=(d["roofline"].get("live_isolated") or d["roofline"]["live"])["kernel_ms_per_spp"]
    print("$name: %.1f Mrays/s %.4f ms/spp in flight %d (%.1f GiB) inline %s fallbacks %s per-frame %s | alone: closest %.4f shadow %.4f shade %.4f" % (d["value"], d["ms_per_spp"], d["config"]["samples_in_flight"], d["config"]["path_state_GB"], d["config"].get("log_inline_entries"), d["config"].get("log_fallbacks"), (d.get("per_frame") or {}).get("mrays_per_s"), k["trace_closest"], k["trace_shadow"], k["shade"]))
except Exception as e:
    print("$name: FAILED", e); print(open("$O/bench_$name.err").read()[-800:])
PY
}
b default; el
b compact --compact-log 1 --per-frame-frames 0; el
b default_again; el
b cfg2_compact_fallback --config 2 --compact-log 1 --steps 4 --per-frame-frames 0; el
b cfg4_32gb_compact --path-state-gb 32 --compact-log 1 --steps 4 --per-frame-frames 0; el
b cfg4_32gb_full --path-state-gb 32 --steps 4 --per-frame-frames 0; el
el all done
