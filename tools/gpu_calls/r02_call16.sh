#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_call16
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -a "passed\|failed\|rror" | tail -6 > $O/pytest_gpu.log
cat $O/pytest_gpu.log
for p in 0 1; do
timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --shade-partition $p > $O/bench_$p.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench_$p.json").read().strip().splitlines()[-1])
print("partition $p:", d["value"], "Mrays/s", d["ms_per_spp"], "ms/spp", d["roofline"]["live"]["kernel_ms_per_spp"])
PY
done
for c in 2 3; do
for p in 0 1; do
timeout 600 python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --shade-partition $p > $O/bench_c${c}_$p.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench_c${c}_$p.json").read().strip().splitlines()[-1])
print("config $c partition $p:", d["value"], "Mrays/s", d["ms_per_spp"], "ms/spp", d["roofline"]["live"]["kernel_ms_per_spp"])
PY
done; done
