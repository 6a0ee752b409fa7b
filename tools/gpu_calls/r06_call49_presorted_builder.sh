#!/bin/bash
# Round 6, call 49: own_bvh.h's presorted small subtrees on the box: the builder alone, the GPU tests that walk own trees, configs 5 / 2 upload lines.
O=gpurun_out/r06_call49; mkdir -p $O
timeout 300 tools/bin/own_bvh_bench 8700000 2>&1 | head -4
timeout 900 python -m pytest tests/test_gpu_device_fold.py tests/test_gpu_headline_parity.py tests/test_gpu_baseline_configs_full_size.py -q -m gpu -x -p no:cacheprovider 2>&1 | grep -aE "passed|failed" | tail -1
for cfg in 5 2; do
  timeout 600 python bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --per-frame-frames 0 --surface-area-fold-steps 0 --cold-job-spp 16 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('cfg $cfg', d['value'], [l[:330] for l in d['cold_job']['trees'] if l.startswith('upload')])"
done
