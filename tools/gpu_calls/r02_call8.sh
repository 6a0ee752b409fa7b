#!/bin/bash
# round 2, GPU call 8: do co-resident half-size persistent grids hide the drain phase of a launch?
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_call8
mkdir -p $O
cd $R
for cfg in "1 0 0" "2 0 13" "2 0 16" "3 0 9" "4 0 7" "4 0 10" "2 16 13" "4 16 7" "4 16 10" "1 16 0"; do
  set -- $cfg
  timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --pipelines $1 --path-state-gb $2 --trace-waves $3 > $O/b.json 2> $O/b.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/b.json").read().strip().splitlines()[-1])
    print("pipelines $1 path-state-gb $2 waves $3:", d["value"], "Mrays/s", d["ms_per_spp"], "ms/spp, state", d["config"]["path_state_GB"], "GB, chunk", d["config"]["chunk_pixels"])
except Exception as e:
    print("pipelines $1 path-state-gb $2 waves $3: FAILED", e); print(open("$O/b.err").read()[-800:])
PY
done > $O/sweep.log 2>&1
cat $O/sweep.log
