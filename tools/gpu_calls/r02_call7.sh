#!/bin/bash
# round 2, GPU call 7: pipelined chunks: parity + sweep of pipes x memory limit
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_call7
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest_gpu.log
cat $O/pytest_gpu.log
for cfg in "1 0" "2 0" "3 0" "4 0" "2 32" "2 16" "4 16" "4 8"; do
  set -- $cfg
  timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --pipelines $1 --path-state-gb $2 > $O/bench_p$1_g$2.json 2> $O/bench_p$1_g$2.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_p$1_g$2.json").read().strip().splitlines()[-1])
    print("pipelines $1 path-state-gb $2:", d["value"], "Mrays/s", d["ms_per_spp"], "ms/spp, state", d["config"]["path_state_GB"], "GB, chunk", d["config"]["chunk_pixels"], d["roofline"]["live"]["kernel_ms_per_spp"])
except Exception as e:
    print("pipelines $1 path-state-gb $2: FAILED", e); print(open("$O/bench_p$1_g$2.err").read()[-800:])
PY
done
