#!/bin/bash
# Round 6, call 50: the upload's host stages after the last two changes (record order on a pool, the choice's inputs measured once): GPU tests that depend on them,
# upload lines of configs 5 / 2 / 4, the driver's command once more.
O=gpurun_out/r06_call50; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_device_fold.py tests/test_gpu_baseline_configs_full_size.py -q -m gpu -x -p no:cacheprovider 2>&1 | grep -aE "passed|failed" | tail -1
for cfg in 5 2; do
  timeout 600 python bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --per-frame-frames 0 --surface-area-fold-steps 0 --cold-job-spp 16 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('cfg $cfg', d['value'], [l[:420] for l in d['cold_job']['trees'] if l.startswith('upload')])"
done
timeout 600 python bench.py > $O/bench_driver_command.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/bench_driver_command.json').read().strip().split('\n')[-1]); c=d['cold_job']
print('bench', d['value'], 'per frame', d['per_frame']['ms_per_frame'], 'parity', d['parity']['bit_identical'], 'adapt', d['adaptation']['seconds_to_adapted'], 'cold', c['upload_s'], c['render_s'], 'setup', d['config']['setup_s'], d['config']['setup_breakdown'], [l[:300] for l in c['trees'] if l.startswith('upload')])"
