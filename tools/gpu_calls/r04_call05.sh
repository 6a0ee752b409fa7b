#!/bin/bash
# Round 4, call 5: loop D as an instance of its own (TAIL), NaN env-map coordinates defined on all three sides: suite + fuzz,
# headline + per-frame on every config, the libm tolerance series on the config-5 stand-in (2 .. 128 spp).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_call05
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
    par = d.get("parity") or {}
    print("$1: %.1f Mrays/s %.4f ms/spp, per-frame %s Mrays/s %s ms | alone: %s | parity: %s" % (
        d["value"], d["ms_per_spp"], pf.get("mrays_per_s"), pf.get("ms_per_frame"), k,
        {x: par.get(x) for x in ("bit_identical", "differing_pixels", "rel_l2", "rel_l2_vs_libm_build")} if par else None))
except Exception as e:
    print("$1: FAILED", e)
PY
}
timeout 900 python -m pytest tests -q -m gpu -x -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror|FAILED" | tail -5 > $O/pytest_gpu.log; el suite: $(tail -1 $O/pytest_gpu.log)
( RT_FUZZ_SEEDS=2000 timeout 600 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -n 32 -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror|Timeout" | tail -3 ) > $O/fuzz_2000_seeds.log 2>&1; el fuzz: $(tail -1 $O/fuzz_2000_seeds.log)
python bench.py --steps 3 --no-cpu-baseline --per-frame-frames 96 > $O/bench_cfg4.json 2>> $O/bench.err; el $(line bench_cfg4)
for cfg in 2 3 5; do
  python bench.py --config $cfg --steps 2 --no-cpu-baseline --per-frame-frames 48 > $O/bench_cfg$cfg.json 2>> $O/bench.err; el $(line bench_cfg$cfg)
done
( timeout 1200 python -X faulthandler tools/libm_tolerance_series.py > $O/libm_tolerance_series_cfg5.json 2> $O/libm_series.err ); el series: $(python -c "
import json; d = json.load(open('$O/libm_tolerance_series_cfg5.json')); print([(p['spp'], '%.2e' % p['rel_l2']) for p in d['series']], d['fitted_slope'], d['crosses_1e_4_at_spp'])")
grep -v amdgpu.ids $O/libm_series.err | tail -5
tail -3 $O/bench.err | grep -v amdgpu.ids
el all done
