#!/bin/bash
# round 3, GPU call 6: chunk mode chosen in the kernel from the live counter -- where to put the threshold
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_call06
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 600 python -m pytest tests -q -m gpu -x -p no:cacheprovider --deselect tests/test_gpu_full_size.py 2>&1 | grep -aE "passed|failed|rror|FAILED|assert" | tail -8 > $O/pytest_gpu.log; el suite: $(tail -1 $O/pytest_gpu.log)
( RT_FUZZ_SEEDS=1000 timeout 300 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -n 32 -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror|Timeout" | tail -3 ) > $O/fuzz_1000_seeds.log 2>&1; el fuzz: $(tail -1 $O/fuzz_1000_seeds.log)
for T in 0 1000000 2000000 3000000 4000000 6000000 10000000; do
  timeout 300 python tools/per_frame_sweep.py --config 4 --frames 32 --settings t$T:0:$T:1:5 2>&1 | tail -1
done > $O/per_frame_threshold_cfg4.log; el pf4; cat $O/per_frame_threshold_cfg4.log
sweep() { cfg=$1; shift; for T in 0 1000000 2000000 3000000 4000000 6000000 10000000; do
  timeout 300 python bench.py --config $cfg --no-cpu-baseline --per-frame-frames 0 --small-launch-paths $T "$@" 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=(d['roofline'].get('live_isolated') or d['roofline']['live'])['kernel_ms_per_spp']
print('cfg $cfg threshold $T $*: %.1f Mrays/s %.4f ms/spp | alone: closest %.4f shadow %.4f shade %.4f' % (d['value'], d['ms_per_spp'], k['trace_closest'], k['trace_shadow'], k['shade']))"
done; }
sweep 4 --steps 3 > $O/threshold_cfg4_headline.log; el head; cat $O/threshold_cfg4_headline.log
sweep 4 --steps 8 --samples-in-flight 8 --samples-per-step 8 > $O/threshold_cfg4_8_in_flight.log; el s8; cat $O/threshold_cfg4_8_in_flight.log
sweep 5 --steps 3 > $O/threshold_cfg5.log; el cfg5; cat $O/threshold_cfg5.log
el all done
