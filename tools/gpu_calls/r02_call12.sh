#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_call12
mkdir -p $O
cd $R
timeout 900 python tools/trace_variants.py --config 4 --slots 128 --spp 128 --variants 10 --tune 32:8:256,32:8:384,32:8:512,32:8:768,32:8:1024,32:8:2048,32:8:4080 > $O/w4_grab.log 2>&1
cat $O/w4_grab.log
