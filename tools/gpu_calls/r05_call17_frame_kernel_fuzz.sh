#!/bin/bash
# Round 5, call 17: the measured choice on HIP events; 2000 fuzz seeds with k_frame among the variants (every fifth seed renders through the stage API
# with RT_OPT_FRAME_KERNEL 1 / 2 / 3); the whole GPU suite.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_call17
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 600 python -m pytest tests/test_gpu_frame_kernel.py -q -m gpu -p no:cacheprovider > $O/pytest_frame_kernel.log 2>&1; el frame kernel tests: $(tail -1 $O/pytest_frame_kernel.log); grep -E "^E " $O/pytest_frame_kernel.log | head -10
RT_FUZZ_SEEDS=2000 timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -n 32 -p no:cacheprovider > $O/fuzz_2000_seeds.log 2>&1; el fuzz: $(tail -1 $O/fuzz_2000_seeds.log); grep -E "^E |^FAILED" $O/fuzz_2000_seeds.log | head -10
timeout 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/pytest_gpu.log 2>&1; el suite: $(grep -aE "passed|failed|rror" $O/pytest_gpu.log | tail -1)
for cfg in 4 2; do timeout 300 python bench.py --config $cfg --per-frame-only --per-frame-frames 96 --moving-camera-frames 0 --frame-kernel 255 > $O/pf_cfg${cfg}_auto.json 2>> $O/bench.err; python -c "
import json; d=json.loads(open('$O/pf_cfg${cfg}_auto.json').read().strip().splitlines()[-1]); print('cfg $cfg measured choice:', d['per_frame']['ms_per_frame'], 'ms per frame')"; done
el all done
