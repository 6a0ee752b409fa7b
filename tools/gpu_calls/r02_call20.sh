#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_call20
mkdir -p $O
cd $R
for ov in 0 1; do
for s in 4 16 128; do
k=$((256 / s)); if [ $k -lt 2 ]; then k=2; fi
timeout 600 python bench.py --steps $k --warmup 1 --samples-per-step $s --samples-in-flight $s --overlap-shadow $ov --no-cpu-baseline > $O/b.json 2> $O/b.err
python - <<PY
import json
d=json.loads(open("$O/b.json").read().strip().splitlines()[-1])
print("overlap $ov samples per step $s:", d["value"], "Mrays/s", d["ms_per_spp"], "ms/spp", d["roofline"]["live"]["kernel_ms_per_spp"])
PY
done; done > $O/overlap.log 2>&1
cat $O/overlap.log
timeout 900 python -m pytest tests/test_gpu_headline_parity.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error" | tail -5
