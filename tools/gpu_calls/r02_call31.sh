#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_call31
mkdir -p $O
cd $R
nproc
( time RT_FUZZ_SEEDS=4000 RT_FUZZ_VARIANT=10 timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -n 24 -p no:cacheprovider 2>&1 | grep -aE "passed|failed|Error|error|Timeout" | tail -5 ) > $O/fuzz_w4.log 2>&1
cat $O/fuzz_w4.log
