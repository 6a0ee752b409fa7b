#!/bin/bash
# FIRST CALL OF THE NEXT ROUND (prepared at the end of round 2, whose GPU budget ended before this could run):
# k_trace_w4<.., DIRECT> (RT_OPT_TRACE_VARIANT 15: the first passing slot is visited next instead of being pushed and popped;
# 8.6 instead of 18.4 pushes per closest-hit ray on the benchmark scene, same node sequence -- DESIGN.md section 2).
#   1. its opt-in tests, 2. the whole GPU suite with the automatic choice switched to it, 3. a fuzz campaign pinned to it,
#   4. A/B against variant 10 on configs 4, 2 and 5.  If all green and faster: make 15 the automatic choice
#   (rt_hip.hip launch_trace: wide_variant default), move its tests into the regular lists, re-collect the evidence.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_call01
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
RT_TEST_EXPERIMENTAL=1 timeout 300 python -m pytest tests/test_gpu_experimental_variants.py -q -m gpu -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror" | tail -3 > $O/pytest_experimental.log; el experimental: $(tail -1 $O/pytest_experimental.log)
RT_TRACE_AUTO_WIDE_VARIANT=15 timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror|FAILED" | tail -6 > $O/pytest_gpu_auto15.log; el auto15 suite: $(tail -1 $O/pytest_gpu_auto15.log)
( time RT_FUZZ_VARIANT=15 RT_FUZZ_SEEDS=2000 timeout 400 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -n 32 -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror|Timeout" | tail -3 ) > $O/fuzz_variant15_2000_seeds.log 2>&1; el fuzz15: $(grep -a "passed\|failed" $O/fuzz_variant15_2000_seeds.log | tail -1)
ab() { name=$1; shift
  timeout 300 python bench.py --no-cpu-baseline "$@" > $O/ab_$name.json 2> $O/ab_$name.err; python - <<PY
import json
try:
    d = json.loads(open("$O/ab_$name.json").read().strip().splitlines()[-1])
    k = (d["roofline"].get("live_isolated") or d["roofline"]["live"])["kernel_ms_per_spp"]
    print("ab $name: %.1f Mrays/s  %.4f ms/spp | alone: closest %.4f shadow %.4f shade %.4f" % (d["value"], d["ms_per_spp"], k["trace_closest"], k["trace_shadow"], k["shade"]))
except Exception as e:
    print("ab $name: FAILED", e)
PY
}
for cfg in 4 2 5; do for v in 10 15 10 15; do ab cfg${cfg}_v${v}_$RANDOM --config $cfg --trace-variant $v | tee -a $O/ab.log; done; done
el all done
