#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_call24
mkdir -p $O
cd $R
for cfg in "1 0" "2 13" "2 16" "4 6" "4 8" "3 8"; do
set -- $cfg
for s in 16; do
timeout 600 python bench.py --steps 4 --warmup 1 --samples-per-step $s --samples-in-flight $s --overlap-shadow 0 --pipelines $1 --trace-waves $2 --no-cpu-baseline > $O/b.json 2> $O/b.err
python - <<PY
import json
d=json.loads(open("$O/b.json").read().strip().splitlines()[-1])
print("pipelines $1 waves/CU $2 in flight $s:", d["value"], "Mrays/s", d["ms_per_spp"], "ms/spp", d["roofline"]["live"]["kernel_ms_per_spp"])
PY
done; done > $O/pipes.log 2>&1
cat $O/pipes.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/bench.py --steps 2 --warmup 1 --samples-per-step 16 --samples-in-flight 16 --overlap-shadow 0 --pipelines 2 --trace-waves 13 --no-cpu-baseline > $O/bt.json 2> $O/bt.err
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
n=$(wc -l < $f)
python $R/tools/kernel_gantt.py $f $((n - 70)) 60 > $O/gantt.log 2>&1
cat $O/gantt.log
