#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_call21
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for ov in 1 0; do
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace_ov$ov -- python $R/bench.py --steps 2 --warmup 1 --samples-per-step 4 --samples-in-flight 4 --overlap-shadow $ov --no-cpu-baseline > $O/b$ov.json 2> $O/b$ov.err
f=$(find $O/trace_ov$ov -name "*kernel_trace.csv" | head -1)
n=$(wc -l < $f)
echo "== overlap $ov: $f ($n rows)"
python $R/tools/kernel_gantt.py $f $((n - 45)) 44
done > $O/gantt.log 2>&1
cat $O/gantt.log
