#!/bin/bash
# round 3, GPU call 19: compact log with k_shade held to 80 VGPRs (6 waves per SIMD): cost in both layouts
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_call19
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 300 python -m pytest tests/test_gpu_headline_parity.py -q -m gpu -x -p no:cacheprovider -k "compact_radiance" > $O/pytest_compact.log 2>&1; el compact test: $(tail -1 $O/pytest_compact.log); grep -aE "^E  " $O/pytest_compact.log | head -6
b() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --per-frame-frames 0 "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1]); k=(d["roofline"].get("live_isolated") or d["roofline"]["live"])["kernel_ms_per_spp"]
    print("$name: %.1f Mrays/s %.4f ms/spp in flight %d (%.1f GiB) inline %s fallbacks %s | alone: closest %.4f shadow %.4f shade %.4f" % (d["value"], d["ms_per_spp"], d["config"]["samples_in_flight"], d["config"]["path_state_GB"], d["config"].get("log_inline_entries"), d["config"].get("log_fallbacks"), k["trace_closest"], k["trace_shadow"], k["shade"]))
except Exception as e:
    print("$name: FAILED", e); print(open("$O/bench_$name.err").read()[-800:])
PY
}
b cfg4_compact --steps 4; el
b cfg4_full --steps 4 --compact-log 0; el
b cfg4_compact_again --steps 4; el
b cfg4_full_again --steps 4 --compact-log 0; el
b cfg5_compact --config 5 --steps 3; el
b cfg5_full --config 5 --steps 3 --compact-log 0; el
b cfg2_compact --config 2 --steps 4; el
b cfg2_full --config 2 --steps 4 --compact-log 0; el
b cfg3 --config 3 --steps 4; el
el all done
