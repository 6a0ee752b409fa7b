#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_call9
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $O/pytest_gpu.log
cat $O/pytest_gpu.log
