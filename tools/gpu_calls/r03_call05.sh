#!/bin/bash
# round 3, GPU call 5: where does chunk mode (static 64-ray chunks, no refill) stop paying? samples in flight 1..64, both modes;
# and the visit-latency micro-benchmark (tools/visit_microbench.hip)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_call05
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 300 tools/bin/visit_mb 0.93 0.85 4096 > $O/visit_microbench.json 2> $O/visit_microbench.err; el visit_mb; cat $O/visit_microbench.json
timeout 300 tools/bin/visit_mb 1.0 1.0 4096 > $O/visit_microbench_all_l1.json 2>> $O/visit_microbench.err; el visit_mb_l1; cat $O/visit_microbench_all_l1.json
timeout 900 python tools/trace_variants.py --config 4 --variants 10 --slots 1,2,4,8,16,32,64 --tune 0x0,0x00800000 --spp 8 > $O/chunk_vs_refill_cfg4.log 2>&1; el cfg4; cat $O/chunk_vs_refill_cfg4.log
timeout 600 python tools/trace_variants.py --config 2 --variants 10 --slots 1,2,4,8,16,32 --tune 0x0,0x00800000 --spp 8 > $O/chunk_vs_refill_cfg2.log 2>&1; el cfg2; cat $O/chunk_vs_refill_cfg2.log
el all done
