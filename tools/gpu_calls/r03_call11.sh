#!/bin/bash
# round 3, GPU call 11: does a quad-cooperative node fetch (one L1 look-up per record instead of four, quarters exchanged through LDS)
# raise the visit rate of the bare chain?  (tools/visit_microbench.hip, 4th argument)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_call11
mkdir -p $O
cd $R
timeout 300 tools/bin/visit_mb 0.93 0.85 4096 1 > $O/visit_microbench_coop.json 2> $O/err.log; cat $O/visit_microbench_coop.json
timeout 300 tools/bin/visit_mb 1.0 1.0 4096 1 > $O/visit_microbench_coop_all_l1.json 2>> $O/err.log; cat $O/visit_microbench_coop_all_l1.json
timeout 300 tools/bin/visit_mb 0.80 0.85 4096 1 > $O/visit_microbench_coop_l1_080.json 2>> $O/err.log; cat $O/visit_microbench_coop_l1_080.json
