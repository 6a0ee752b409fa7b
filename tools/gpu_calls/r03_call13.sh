#!/bin/bash
# round 3, GPU call 13: residency of the persistent trace grids re-swept on round 3's kernels (the bare chain peaks at 20 waves per CU)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_call13
mkdir -p $O
cd $R
for W in 16 18 20 22 24 26; do
  timeout 300 python bench.py --no-cpu-baseline --per-frame-frames 0 --steps 3 --trace-waves $W 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['live_isolated']['kernel_ms_per_spp']
print('waves per CU $W: %.1f Mrays/s %.4f ms/spp | alone: closest %.4f shadow %.4f shade %.4f' % (d['value'], d['ms_per_spp'], k['trace_closest'], k['trace_shadow'], k['shade']))"
done | tee $O/trace_waves_sweep.log
