#!/bin/bash
# Round 3, call 29: the GPU suite after the host-side changes (threaded wide-tree records), incl. the two-rank bench test that
# now also asks RCCL for a communicator of two ranks on ONE device (refused) and must fall back to the gloo gather by agreement.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_call29
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 600 python -m pytest tests/test_gpu_bench_scene.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -15 > $O/pytest_bench_scene.log; el bench tests: $(tail -1 $O/pytest_bench_scene.log)
timeout 300 python bench.py --gpus 2 --debug-shared-gpu --debug-try-rccl --config 2 --steps 1 --samples-per-step 16 > $O/bench_2rank_try_rccl.json 2> $O/bench_2rank_try_rccl.err; el $(python -c "
import json; d=json.loads(open('$O/bench_2rank_try_rccl.json').read().strip().splitlines()[-1]); print(d['value'], d['gather'], d['parity'])" 2>&1 | tail -1 | cut -c1-700)
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror|FAILED" | tail -4 > $O/pytest_gpu.log; el suite: $(tail -1 $O/pytest_gpu.log)
el all done
