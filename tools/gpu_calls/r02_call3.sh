#!/bin/bash
# round 2, GPU call 3: k_trace_w4 parity + first timing
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_call3
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest_gpu.log
cat $O/pytest_gpu.log
timeout 900 python tools/trace_variants.py --config 4 --slots 128 --spp 128 --variants 8,10,11 --tune 32:8,24:8,40:12,16:8,48:16 > $O/variants_cfg4.log 2>&1
cat $O/variants_cfg4.log
