#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_call32
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_headline_parity.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error|Timeout" | tail -5
for rep in 1 2; do
for s in 128 16; do
k=$((512 / s)); if [ $k -gt 8 ]; then k=8; fi
timeout 300 python bench.py --steps $k --warmup 1 --samples-per-step $s --samples-in-flight $s --no-cpu-baseline > $O/b.json 2> $O/b.err
python - <<PY
import json
d=json.loads(open("$O/b.json").read().strip().splitlines()[-1])
print("rep $rep in flight $s:", d["value"], "Mrays/s", d["ms_per_spp"], "ms/spp", d["roofline"]["live"]["kernel_ms_per_spp"])
PY
done; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o stats -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/stats.log 2>&1
cut -c1-40 $O/stats/stats_kernel_stats.csv | head -3; awk -F'","' '{print substr($1,1,30), $2, $4}' $O/stats/stats_kernel_stats.csv | head -8
find $O/stats -name "*.csv" -size +3M -delete
