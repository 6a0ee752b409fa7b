#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_call23
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for pp in 2 4; do
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace_p$pp -- python $R/bench.py --steps 2 --warmup 1 --samples-per-step 16 --samples-in-flight 16 --overlap-shadow 0 --pipelines $pp --no-cpu-baseline > $O/b$pp.json 2> $O/b$pp.err
f=$(find $O/trace_p$pp -name "*kernel_trace.csv" | head -1)
n=$(wc -l < $f)
echo "== pipelines $pp: $f ($n rows)"
python - <<PY
import json
d=json.loads(open("$O/b$pp.json").read().strip().splitlines()[-1])
print(d["value"], "Mrays/s", d["ms_per_spp"], "ms/spp", d["config"].get("pipelines"), d["roofline"]["live"]["kernel_ms_per_spp"])
PY
python $R/tools/kernel_gantt.py $f $((n - 90)) 89
done > $O/gantt.log 2>&1
cat $O/gantt.log
