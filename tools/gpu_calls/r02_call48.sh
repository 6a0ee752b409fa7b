#!/bin/bash
# Round 2, call 48 (last): smoke() and the GPU suite on the tree as it is committed at the end of the round.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_closing_sanity
mkdir -p $O
cd $R
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -a "smoke" | tail -1 | tee $O/smoke.log
timeout 70 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror" | tail -3 | tee $O/pytest_gpu.log
