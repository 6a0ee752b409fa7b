#!/bin/bash
# round 2, GPU call 2: occupancy sweep of k_trace2 (latency- or throughput-bound?)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_call2
mkdir -p $O
cd $R
timeout 900 python tools/trace_variants.py --config 4 --slots 128 --spp 128 --variants 8 --tune 32:8 --waves 8,12,16,20,24,28,32 > $O/waves_v8.log 2>&1
cat $O/waves_v8.log
timeout 900 python tools/trace_variants.py --config 4 --slots 128 --spp 128 --variants 5 --waves 8,16,24,32 > $O/waves_v5.log 2>&1
cat $O/waves_v5.log
