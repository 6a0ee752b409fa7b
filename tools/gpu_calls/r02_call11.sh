#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_call11
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $O/pytest_gpu.log
cat $O/pytest_gpu.log
timeout 900 python tools/trace_variants.py --config 4 --slots 128 --spp 128 --variants 10 --tune 32:8,32:8:64,32:8:32,32:8:256,32:8:16 > $O/w4_grab.log 2>&1
cat $O/w4_grab.log
