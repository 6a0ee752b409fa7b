#!/bin/bash
# Round 4, call 2: shadow rays read their own copy of the trace-triangle records in their tree's leaf order
# (RT_CTX_OPT_SHADOW_TRIANGLE_COPY).  Suite (default = automatic choice with a 10 % threshold + the copy), then the A/B with the
# own tree forced on configs 3 / 4 / 2 / 5, copy off / on.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_call02
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
line() { python - <<PY
import json
try:
    d = json.loads(open("$O/$1.json").read().strip().splitlines()[-1])
    k = (d["roofline"].get("live_isolated") or d["roofline"]["live"])["kernel_ms_per_spp"]
    print("$1: %.1f Mrays/s %.4f ms/spp, setup %s s | alone: %s | %s" % (d["value"], d["ms_per_spp"], d["config"].get("setup_s"), k, d["config"].get("trees")))
except Exception as e:
    print("$1: FAILED", e)
PY
}
timeout 900 python -m pytest tests -q -m gpu -x -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror|FAILED" | tail -5 > $O/pytest_gpu.log; el suite: $(tail -1 $O/pytest_gpu.log)
for cfg in 3 4 2 5; do
  for c in 0 1; do
    python bench.py --config $cfg --steps 2 --no-cpu-baseline --per-frame-frames 0 --shadow-tree 2 --shadow-tris-copy $c > $O/bench_cfg${cfg}_own_copy$c.json 2>> $O/bench.err; el $(line bench_cfg${cfg}_own_copy$c)
  done
done
python bench.py --config 3 --steps 2 --no-cpu-baseline --per-frame-frames 0 > $O/bench_cfg3_default.json 2>> $O/bench.err; el $(line bench_cfg3_default)
tail -5 $O/bench.err
el all done
