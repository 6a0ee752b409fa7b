#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_call17
mkdir -p $O
cd $R
for s in 4 8 16 32 64 128; do
k=$((256 / s)); if [ $k -lt 2 ]; then k=2; fi
timeout 600 python bench.py --steps $k --warmup 1 --samples-per-step $s --samples-in-flight $s --no-cpu-baseline > $O/b.json 2> $O/b.err
python - <<PY
import json
d=json.loads(open("$O/b.json").read().strip().splitlines()[-1])
print("samples per step $s:", d["value"], "Mrays/s", d["ms_per_spp"], "ms/spp, in flight", d["config"]["samples_in_flight"], d["roofline"]["live"]["kernel_ms_per_spp"], "rays/launch", d["roofline"]["live"]["rays_per_launch"], "ms/launch", d["roofline"]["live"]["avg_launch_ms"])
PY
done > $O/batch_size.log 2>&1
cat $O/batch_size.log
