#!/bin/bash
# Round 4, call 1: the trees of the backend's own (own_bvh.h + tree_select.h).  Suite + fuzz with the shadow tree chosen
# automatically (default), then the A/B on every config: shadow rays on the shared tree (0) / automatic (1) / own forced (2) /
# own with the surface-area metric (3); closest-hit rays on an own tree (tolerance mode) with the parity leg to count pixels.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_call01
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
    print("$1: %.1f Mrays/s %.4f ms/spp, per-frame %s, setup %s s | alone: %s | parity: %s | %s" % (
        d["value"], d["ms_per_spp"], pf.get("mrays_per_s"), d["config"].get("setup_s"), k,
        {x: par.get(x) for x in ("bit_identical", "differing_pixels", "rel_l2")} if par else None, d["config"].get("trees")))
except Exception as e:
    print("$1: FAILED", e)
PY
}
nproc; python -c "import os; print(os.cpu_count())"
timeout 900 python -m pytest tests -q -m gpu -x -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror|FAILED" | tail -5 > $O/pytest_gpu.log; el suite: $(tail -1 $O/pytest_gpu.log)
( RT_FUZZ_SEEDS=1500 timeout 600 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -n 32 -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror|Timeout" | tail -3 ) > $O/fuzz_1500_seeds.log 2>&1; el fuzz: $(tail -1 $O/fuzz_1500_seeds.log)
for t in 0 1 2 3; do
  python bench.py --steps 3 --no-cpu-baseline --per-frame-frames 24 --shadow-tree $t > $O/bench_cfg4_shadow$t.json 2>> $O/bench.err; el $(line bench_cfg4_shadow$t)
done
for cfg in 2 3 5; do
  for t in 0 1 2; do
    python bench.py --config $cfg --steps 2 --no-cpu-baseline --per-frame-frames 0 --shadow-tree $t > $O/bench_cfg${cfg}_shadow$t.json 2>> $O/bench.err; el $(line bench_cfg${cfg}_shadow$t)
  done
done
# tolerance mode: closest-hit rays on an own tree; the CPU leg renders the reference's frame so that parity counts the pixels
python bench.py --steps 3 --per-frame-frames 0 --closest-tree 2 --cpu-seconds 8 > $O/bench_cfg4_closest2.json 2>> $O/bench.err; el $(line bench_cfg4_closest2)
python bench.py --config 2 --steps 2 --per-frame-frames 0 --closest-tree 2 --cpu-seconds 8 > $O/bench_cfg2_closest2.json 2>> $O/bench.err; el $(line bench_cfg2_closest2)
tail -5 $O/bench.err
el all done
