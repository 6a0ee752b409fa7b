#!/bin/bash
# round 3, GPU call 2: parity of the pruned library with the 412-byte path state (no 1/dir queues, 12-byte log entries),
# headline, first per-frame numbers (the reference's call pattern) and a sweep of the small-launch settings
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_call02
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 900 python -m pytest tests -q -m gpu -x -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror|FAILED|assert" | tail -12 > $O/pytest_gpu.log; el suite: $(tail -1 $O/pytest_gpu.log)
( RT_FUZZ_SEEDS=1500 timeout 400 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -n 32 -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror|Timeout" | tail -3 ) > $O/fuzz_1500_seeds.log 2>&1; el fuzz: $(tail -1 $O/fuzz_1500_seeds.log)
timeout 400 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; el bench: $(python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); k=(d['roofline'].get('live_isolated') or d['roofline']['live'])['kernel_ms_per_spp']
print(d['value'], 'Mrays/s', d['ms_per_spp'], 'ms/spp alone:', k, 'per_frame:', d['per_frame']['mrays_per_s'], d['per_frame']['ms_per_frame'], 'GB', d['config']['path_state_GB'])" 2>&1 | tail -1)
timeout 600 python tools/per_frame_sweep.py --config 4 --frames 32 --settings \
  base_v1rule:0:2000000:1:5 base_no_overlap:0:2000000:0:5 w4_full_grid:0:0:1:5 \
  w4_rpl1:0x01000000:0:1:5 w4_rpl2:0x02000000:0:1:5 w4_rpl4:0x04000000:0:1:5 w4_rpl8:0x08000000:0:1:5 w4_rpl16:0x10000000:0:1:5 w4_rpl32:0x20000000:0:1:5 \
  w4_rpl4_grab4:0x04040000:0:1:5 w4_rpl8_grab4:0x08040000:0:1:5 w4_rpl4_no_overlap:0x04000000:0:0:5 \
  v1_always:0:4000000000:1:5 k2:0:0:1:8 > $O/per_frame_sweep_cfg4.log 2>&1; el sweep4; cat $O/per_frame_sweep_cfg4.log
timeout 300 python tools/per_frame_sweep.py --config 2 --frames 32 --settings \
  base_v1rule:0:2000000:1:5 w4_full_grid:0:0:1:5 w4_rpl2:0x02000000:0:1:5 w4_rpl4:0x04000000:0:1:5 w4_rpl8:0x08000000:0:1:5 w4_rpl16:0x10000000:0:1:5 \
  > $O/per_frame_sweep_cfg2.log 2>&1; el sweep2; cat $O/per_frame_sweep_cfg2.log
el all done
