#!/bin/bash
# Round 5, call 8: the presented image's device-to-host copy against the kernels that run meanwhile (tools/d2h_copy_probe.hip).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_call08
mkdir -p $O
cd $R
echo "--- default"; timeout 120 tools/bin/d2h_copy_probe 2>&1 | tee $O/probe_default.log
echo "--- HSA_ENABLE_SDMA=0"; HSA_ENABLE_SDMA=0 timeout 120 tools/bin/d2h_copy_probe 2>&1 | tee $O/probe_sdma0.log
( cd /tmp && export TMPDIR=/tmp && timeout 120 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/trace -o probe -- $R/tools/bin/d2h_copy_probe > $O/probe_traced.log 2>&1 )
echo "--- kernels and copies seen by rocprofv3"; cat $O/trace/*kernel_stats.csv 2>/dev/null | cut -c1-160 | head -8; cat $O/trace/*memory_copy_stats.csv 2>/dev/null | cut -c1-200 | head -8
find $O/trace -name "*.csv" -size +1M -delete
