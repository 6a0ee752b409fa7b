#!/bin/bash
# round 3, GPU call 21: k_shade's compact-log code in instances of its own: the default instances as before?
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_call21
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest_gpu_full.log 2>&1; el suite: $(grep -aE "passed|failed" $O/pytest_gpu_full.log | tail -1); grep -aE "^FAILED|^ERROR|^E  " $O/pytest_gpu_full.log | head
( RT_FUZZ_SEEDS=1500 timeout 600 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -n 32 -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror|Timeout" | tail -3 ) > $O/fuzz_1500_seeds.log 2>&1; el fuzz: $(tail -1 $O/fuzz_1500_seeds.log)
b() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --per-frame-frames 0 "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1]); k=(d["roofline"].get("live_isolated") or d["roofline"]["live"])["kernel_ms_per_spp"]
    print("$name: %.1f Mrays/s %.4f ms/spp in flight %d (%.1f GiB) inline %s | alone: closest %.4f shadow %.4f shade %.4f" % (d["value"], d["ms_per_spp"], d["config"]["samples_in_flight"], d["config"]["path_state_GB"], d["config"].get("log_inline_entries"), k["trace_closest"], k["trace_shadow"], k["shade"]))
except Exception as e:
    print("$name: FAILED", e); print(open("$O/bench_$name.err").read()[-800:])
PY
}
b default; b compact --compact-log 1; b default_again; b compact_again --compact-log 1; b default_3
el all done
