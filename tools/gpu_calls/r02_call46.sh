#!/bin/bash
# Round 2, call 46: the LEAN instances of k_trace_w4 (results written when found, 8 registers fewer): variant 15 = 12-entry LDS
# stack (26 waves per CU like variant 10: what LEAN itself costs), variant 16 = 10-entry stack at <= 64 VGPRs (31 waves per CU).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_call46
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 600 python -m pytest tests -q -m gpu -x -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror" | tail -3 > $O/pytest_gpu_default.log; el default suite: $(tail -1 $O/pytest_gpu_default.log)
RT_TRACE_AUTO_WIDE_VARIANT=16 timeout 600 python -m pytest tests -q -m gpu -x -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror" | tail -3 > $O/pytest_gpu_auto16.log; el auto16 suite: $(tail -1 $O/pytest_gpu_auto16.log)
( time RT_FUZZ_VARIANT=16 RT_FUZZ_SEEDS=1500 timeout 400 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -n 32 -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror|Timeout" | tail -3 ) > $O/fuzz_variant16_1500_seeds.log 2>&1; el fuzz16: $(grep -a "passed\|failed" $O/fuzz_variant16_1500_seeds.log | tail -1)
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
ab v10 --trace-variant 10 | tee -a $O/ab.log
ab v15_lean12 --trace-variant 15 | tee -a $O/ab.log
ab v16_lean10 --trace-variant 16 | tee -a $O/ab.log
ab v14_stack10 --trace-variant 14 | tee -a $O/ab.log
ab v16_waves28 --trace-variant 16 --trace-waves 28 | tee -a $O/ab.log
ab v16_waves30 --trace-variant 16 --trace-waves 30 | tee -a $O/ab.log
ab v10_again --trace-variant 10 | tee -a $O/ab.log
ab cfg5_v10 --config 5 --trace-variant 10 | tee -a $O/ab.log
ab cfg5_v16 --config 5 --trace-variant 16 | tee -a $O/ab.log
el all done
