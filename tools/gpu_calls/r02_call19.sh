#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_call19
mkdir -p $O
cd $R
for t in 0x0F000820 0x01000820 0x02000820 0x04000820 0x08000820; do
echo "== tune $t"
timeout 300 python tools/launch_timeline.py --in-flight 4,128 --tune $t 2>&1 | grep -E "samples in flight|all bounces|bounce 1:|bounce 7:"
done > $O/taper.log 2>&1
cat $O/taper.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
