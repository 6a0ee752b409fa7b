#!/bin/bash
# round 2, GPU call 6: new parity tests (headline shapes, spill, chunking) + path-state sweep
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_call6
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest_gpu.log
cat $O/pytest_gpu.log
for g in 0 32 16 8; do
  timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --path-state-gb $g > $O/bench_state_$g.json 2> $O/bench_state_$g.err
  python - <<PY
import json
d=json.loads(open("$O/bench_state_$g.json").read().strip().splitlines()[-1])
print("path-state-gb $g:", d["value"], "Mrays/s", d["ms_per_spp"], "ms/spp, state", d["config"]["path_state_GB"], "GB, chunk", d["config"]["chunk_pixels"], d["roofline"]["live"]["kernel_ms_per_spp"])
PY
done
