#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_call33
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error|Timeout|assert" | tail -8
for rep in 1 2; do
timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/b.json 2> $O/b.err
python - <<PY
import json
d=json.loads(open("$O/b.json").read().strip().splitlines()[-1])
print("rep $rep:", d["value"], "Mrays/s", d["ms_per_spp"], "ms/spp", d["roofline"]["live"]["kernel_ms_per_spp"], d["roofline"].get("live_isolated", {}).get("kernel_ms_per_spp"))
PY
done
