#!/bin/bash
# Round 4, call 25: the mid-sample test on an open scene, the whole suite again.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_call25
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror|FAILED|assert" | tail -8 | tee $O/pytest_gpu.log
