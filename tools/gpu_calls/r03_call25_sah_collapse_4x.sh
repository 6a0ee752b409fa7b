#!/bin/bash
# Round 3, call 25: call 24 again with FOUR conditional exchanges in k_trace_w4 instead of five (the slots of a record are
# now stored where that network can produce every order the record's shape asks for; call 24 showed that the fifth
# exchange cost 4 % of the closest-hit kernel, most of what the SAH collapse gained): suite + fuzz, the A/B on the headline
# workload against the two-level collapse (RT_CTX_OPT_WIDE_BVH = 2), the other configs, the bare visit chain.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_call25
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
line() { python - <<PY
import json
try:
    d = json.loads(open("$O/$1.json").read().strip().splitlines()[-1])
    pf = d.get("per_frame") or {}
    k = (d["roofline"].get("live_isolated") or d["roofline"]["live"])["kernel_ms_per_spp"]
    print("$1: %.1f Mrays/s %.4f ms/spp, per-frame %s Mrays/s, steps/ray %s | alone: %s" % (
        d["value"], d["ms_per_spp"], pf.get("mrays_per_s"), (d["roofline"].get("latency_ceiling") or {}).get("steps_per_ray"), k))
except Exception as e:
    print("$1: FAILED", e)
PY
}
timeout 900 python -m pytest tests -q -m gpu -x -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror|FAILED" | tail -5 > $O/pytest_gpu.log; el suite: $(tail -1 $O/pytest_gpu.log)
( RT_FUZZ_SEEDS=1500 timeout 600 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -n 32 -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror|Timeout" | tail -3 ) > $O/fuzz_1500_seeds.log 2>&1; el fuzz: $(tail -1 $O/fuzz_1500_seeds.log)
for c in 2 1; do
  python bench.py --steps 3 --no-cpu-baseline --per-frame-frames 24 --wide-collapse $c > $O/bench_cfg4_collapse$c.json 2>> $O/bench.err; el $(line bench_cfg4_collapse$c)
done
for c in 1; do
  python bench.py --config 5 --steps 2 --no-cpu-baseline --per-frame-frames 0 --wide-collapse $c > $O/bench_cfg5_collapse$c.json 2>> $O/bench.err; el $(line bench_cfg5_collapse$c)
done
for c in 1; do
  python bench.py --config 2 --steps 3 --no-cpu-baseline --per-frame-frames 0 --wide-collapse $c > $O/bench_cfg2_collapse$c.json 2>> $O/bench.err; el $(line bench_cfg2_collapse$c)
  python bench.py --config 3 --steps 3 --no-cpu-baseline --per-frame-frames 0 --wide-collapse $c > $O/bench_cfg3_collapse$c.json 2>> $O/bench.err; el $(line bench_cfg3_collapse$c)
done
timeout 300 tools/bin/visit_mb 0.93 0.85 4096 > $O/visit_microbench.json 2> $O/visit_microbench.err; el visit_mb; cat $O/visit_microbench.json
tail -3 $O/bench.err
el all done
