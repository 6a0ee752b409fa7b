#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_call37
mkdir -p $O
cd $R
for t in 0x0820 0x0818 0x0828 0x0420 0x1020 0x0830 0x0C28 0x0620; do
timeout 300 python bench.py --steps 3 --warmup 1 --trace-tune $t --no-cpu-baseline > $O/b.json 2> $O/b.err
python - <<PY
import json
d=json.loads(open("$O/b.json").read().strip().splitlines()[-1])
print("tune $t:", d["value"], "Mrays/s", d["ms_per_spp"], "ms/spp", d["roofline"].get("live_isolated", {}).get("kernel_ms_per_spp"))
PY
done > $O/tune.log 2>&1
cat $O/tune.log
