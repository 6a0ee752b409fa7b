#!/bin/bash
# Round 3, call 26: issue rates of the opcodes loop C of k_trace_w4 is made of (tools/issue_microbench.hip, extended): which of its
# selects, conversions and tests have a cheaper form on gfx950?
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_call26
mkdir -p $O
cd $R
timeout 300 tools/bin/issue_mb valu > $O/issue_microbench_valu.log 2>&1
cat $O/issue_microbench_valu.log | cut -c1-175
