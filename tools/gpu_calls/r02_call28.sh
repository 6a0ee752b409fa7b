#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_call28
mkdir -p $O
cd $R
for t in 0x3F000820 0x18000820; do
echo "== tune $t"
timeout 300 python tools/launch_timeline.py --in-flight 4,128 --tune $t 2>&1 | grep -E "samples in flight|all bounces|bounce 1:|bounce 7:|waves gone"
done > $O/hist.log 2>&1
cat $O/hist.log
