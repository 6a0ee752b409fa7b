#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_call25
mkdir -p $O
cd $R
for rep in 1 2 3; do
for ov in 0 1; do
for s in 128; do
timeout 600 python bench.py --steps 4 --warmup 1 --samples-per-step $s --samples-in-flight $s --overlap-shadow $ov --no-cpu-baseline > $O/b.json 2> $O/b.err
python - <<PY
import json
d=json.loads(open("$O/b.json").read().strip().splitlines()[-1])
print("rep $rep overlap $ov samples per step $s:", d["value"], "Mrays/s", d["ms_per_spp"], "ms/spp", d["roofline"]["live"]["kernel_ms_per_spp"])
PY
done; done; done > $O/overlap.log 2>&1
cat $O/overlap.log
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error" | tail -5
