#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_call26
mkdir -p $O
cd $R
timeout 500 python tools/launch_timeline.py --in-flight 4,128 > $O/timeline.log 2>&1
cat $O/timeline.log
