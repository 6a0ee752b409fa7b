#!/bin/bash
# Round 4, call 8: compact log sub-pools with stealing (suite + fuzz + A/B incl. config 5), the presented image back on the runtime's
# copy but on a low-priority stream, the visit micro-benchmark with exactly the production kernel's LDS footprint.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_call08
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
line() { python - <<PY
import json
try:
    d = json.loads(open("$O/$1.json").read().strip().splitlines()[-1])
    k = (d["roofline"].get("live_isolated") or d["roofline"]["live"])["kernel_ms_per_spp"]
    print("$1: %.1f Mrays/s %.4f ms/spp, %s GiB, fallbacks %s | alone: %s" % (d["value"], d["ms_per_spp"], d["config"]["path_state_GB"], d["config"]["log_fallbacks"], k))
except Exception as e:
    print("$1: FAILED", e)
PY
}
timeout 900 python -m pytest tests -q -m gpu -x -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror|FAILED" | tail -5 > $O/pytest_gpu.log; el suite: $(tail -1 $O/pytest_gpu.log)
( RT_FUZZ_SEEDS=1500 timeout 600 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -n 32 -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror|Timeout" | tail -3 ) > $O/fuzz_1500_seeds.log 2>&1; el fuzz: $(tail -1 $O/fuzz_1500_seeds.log)
for cfg in 4 2 3; do
  python bench.py --config $cfg --steps 1 --no-cpu-baseline --per-frame-frames 96 --per-frame-only > $O/pf_cfg$cfg.json 2>> $O/bench.err; el per-frame cfg $cfg: $(python -c "
import json; d=json.loads(open('$O/pf_cfg$cfg.json').read().strip().splitlines()[-1]); print(d['per_frame']['mrays_per_s'], d['per_frame']['ms_per_frame'])")
done
for c in 0 1 0 1; do
  python bench.py --steps 3 --no-cpu-baseline --per-frame-frames 0 --compact-log $c > $O/bench_cfg4_compact$c.json 2>> $O/bench.err; el $(line bench_cfg4_compact$c)
done
python bench.py --config 5 --steps 2 --no-cpu-baseline --per-frame-frames 0 --compact-log 1 > $O/bench_cfg5_compact1.json 2>> $O/bench.err; el $(line bench_cfg5_compact1)
python bench.py --config 4 --path-state-gb 32 --steps 3 --no-cpu-baseline --per-frame-frames 0 > $O/bench_cfg4_32GiB.json 2>> $O/bench.err; el $(line bench_cfg4_32GiB)
timeout 300 tools/bin/visit_mb 0.93 0.87 4096 > $O/visit_microbench.json 2> $O/visit_microbench.err; el visit_mb: $(python -c "
import json; d=json.load(open('$O/visit_microbench.json')); print([(r['kernel'][:3], r['waves_per_cu'], round(r['gvisits_per_s'],1)) for r in d['runs']])")
tail -3 $O/bench.err | grep -v amdgpu.ids
el all done
