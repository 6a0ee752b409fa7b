#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_call14
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -a "passed\|failed\|rror" | tail -6 > $O/pytest_gpu.log
cat $O/pytest_gpu.log
timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["value"], "Mrays/s", d["ms_per_spp"], "ms/spp", d["roofline"]["live"]["kernel_ms_per_spp"])
PY
