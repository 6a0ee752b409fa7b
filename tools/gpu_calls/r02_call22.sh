#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_call22
mkdir -p $O
cd $R
for t in 0x00000820 0x10000820 0x30000820; do
for ov in 0 1; do
for s in 4 128; do
k=$((256 / s)); if [ $k -lt 2 ]; then k=2; fi
timeout 600 python bench.py --steps $k --warmup 1 --samples-per-step $s --samples-in-flight $s --overlap-shadow $ov --trace-tune $t --no-cpu-baseline > $O/b.json 2> $O/b.err
python - <<PY
import json
d=json.loads(open("$O/b.json").read().strip().splitlines()[-1])
print("tune $t overlap $ov samples per step $s:", d["value"], "Mrays/s", d["ms_per_spp"], "ms/spp", d["roofline"]["live"]["kernel_ms_per_spp"])
PY
done; done; done > $O/prio.log 2>&1
cat $O/prio.log
