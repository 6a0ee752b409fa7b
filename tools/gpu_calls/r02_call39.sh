#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_call39
mkdir -p $O
cd $R
( time RT_FUZZ_SEEDS=1500 timeout 600 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -n 32 -p no:cacheprovider 2>&1 | grep -aE "passed|failed|Error|error|Timeout" | tail -5 ) > $O/fuzz_ext.log 2>&1
cat $O/fuzz_ext.log
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error|Timeout" | tail -3
