#!/bin/bash
# round 3, GPU call 14: the remaining launch parameters re-swept on round 3's kernels (loop thresholds, rays per hand-out, samples in flight)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_call14
mkdir -p $O
cd $R
run() { name=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --per-frame-frames 0 --steps 4 "$@" 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['live_isolated']['kernel_ms_per_spp']
print('$name: %.1f Mrays/s %.4f ms/spp in flight %d | alone: closest %.4f shadow %.4f shade %.4f' % (d['value'], d['ms_per_spp'], d['config']['samples_in_flight'], k['trace_closest'], k['trace_shadow'], k['shade']))"; }
{
run default
run q24_8 --trace-tune 0x0818
run q40_8 --trace-tune 0x0828
run q32_4 --trace-tune 0x0420
run q32_12 --trace-tune 0x0c20
run q48_16 --trace-tune 0x1030
run grab256 --trace-tune 0x100000
run grab1024 --trace-tune 0x400000
run grab2032 --trace-tune 0x7f0000
run inflight96 --samples-in-flight 96 --samples-per-step 96
run inflight160 --samples-in-flight 160 --samples-per-step 160
run partition1 --shade-partition 1
run partition2 --shade-partition 2
run default_again
} | tee $O/launch_parameter_sweep.log
