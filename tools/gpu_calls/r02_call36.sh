#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_call36
mkdir -p $O
cd $R
( time RT_FUZZ_SEEDS=12000 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -n 32 -p no:cacheprovider 2>&1 | grep -aE "passed|failed|Error|error|Timeout" | tail -5 ) > $O/fuzz_all.log 2>&1
cat $O/fuzz_all.log
bash tools/collect_evidence.sh r02_final
