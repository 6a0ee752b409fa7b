#!/bin/bash
# Round 2, call 49 (the round's last GPU seconds): k_trace_w4's DIRECT instance (variant 15: the first passing slot is visited
# next instead of being pushed and popped) -- the tests that pin it, then one A/B pair.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_call49
mkdir -p $O
cd $R
( RT_FUZZ_VARIANT=15 timeout 40 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline_parity.py tests/test_gpu_fuzz.py -q -m gpu -x -p no:cacheprovider -k "15- or -15 or deep_tree or fuzz" 2>&1 | grep -aE "passed|failed|rror" | tail -3 ) | tee $O/pytest_variant15.log
for v in 10 15; do timeout 20 python bench.py --no-cpu-baseline --steps 4 --trace-variant $v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['live_isolated']['kernel_ms_per_spp']
print('ab v$v: %.1f Mrays/s | alone: closest %.4f shadow %.4f shade %.4f' % (d['value'], k['trace_closest'], k['trace_shadow'], k['shade']))" | tee -a $O/ab.log; done
