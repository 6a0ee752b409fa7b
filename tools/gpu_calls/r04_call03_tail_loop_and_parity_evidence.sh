#!/bin/bash
# Round 4, call 3: k_trace_w4's loop D (the fused tail pass, RT_OPT_TRACE_TAIL_LANES) -- suite + fuzz with it on (default 16),
# the sweep on the headline + per-frame legs; the new parity evidence: config 1 on its real asset, the libm-build tolerance test,
# the tolerance series on the config-5 stand-in (tools/libm_tolerance_series.py), bench --config 1 and --config 5 lines.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_call03
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
for t in 0 8 16 32 64; do
  python bench.py --steps 3 --no-cpu-baseline --per-frame-frames 48 --tail-lanes $t > $O/bench_cfg4_tail$t.json 2>> $O/bench.err; el $(line bench_cfg4_tail$t)
done
for cfg in 2 5; do
  for t in 0 16; do
    python bench.py --config $cfg --steps 2 --no-cpu-baseline --per-frame-frames 24 --tail-lanes $t > $O/bench_cfg${cfg}_tail$t.json 2>> $O/bench.err; el $(line bench_cfg${cfg}_tail$t)
  done
done
python bench.py --config 1 --steps 64 --warmup 4 > $O/bench_cfg1.json 2>> $O/bench.err; el $(line bench_cfg1)
python -c "
import json; d = json.loads(open('$O/bench_cfg1.json').read().strip().splitlines()[-1]); print(json.dumps(d['parity'].get('config_1_sample_0')))"
( timeout 900 python tools/libm_tolerance_series.py > $O/libm_tolerance_series_cfg5.json 2> $O/libm_series.err ); el series: $(python -c "
import json; d = json.load(open('$O/libm_tolerance_series_cfg5.json')); print([(p['spp'], '%.2e' % p['rel_l2']) for p in d['series']], d['fitted_slope'], d['crosses_1e_4_at_spp'])")
python bench.py --config 5 --steps 2 --per-frame-frames 0 --cpu-seconds 8 > $O/bench_cfg5.json 2>> $O/bench.err; el $(line bench_cfg5)
python -c "
import json; d = json.loads(open('$O/bench_cfg5.json').read().strip().splitlines()[-1]); print(json.dumps(d['parity'].get('rel_l2_vs_libm_build_series')))"
tail -5 $O/bench.err | grep -v amdgpu.ids
el all done
