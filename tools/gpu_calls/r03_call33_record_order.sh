#!/bin/bash
# Round 3, call 33: the order of the wide records in memory (host-side only; the kernels are untouched): depth-first (1, shipped),
# a record's interior children next to each other (3), breadth-first clusters of 8 records (4).  Same box, twice each.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_call33
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
line() { python - <<PY
import json
try:
    d = json.loads(open("$O/$1.json").read().strip().splitlines()[-1])
    k = (d["roofline"].get("live_isolated") or d["roofline"]["live"])["kernel_ms_per_spp"]
    print("$1: %.1f Mrays/s %.4f ms/spp | alone: %s" % (d["value"], d["ms_per_spp"], k))
except Exception as e:
    print("$1: FAILED", e)
PY
}
B="--steps 3 --no-cpu-baseline --per-frame-frames 0"
for rep in a b; do for c in 1 3 4; do
  python bench.py $B --wide-collapse $c > $O/bench_order${c}_$rep.json 2>> $O/bench.err; el $(line bench_order${c}_$rep)
done; done
python bench.py --config 5 --steps 1 --no-cpu-baseline --per-frame-frames 0 --wide-collapse 1 > $O/bench_cfg5_order1.json 2>> $O/bench.err; el $(line bench_cfg5_order1)
python bench.py --config 5 --steps 1 --no-cpu-baseline --per-frame-frames 0 --wide-collapse 4 > $O/bench_cfg5_order4.json 2>> $O/bench.err; el $(line bench_cfg5_order4)
tail -2 $O/bench.err
el all done
