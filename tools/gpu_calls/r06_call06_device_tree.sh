#!/bin/bash
# Round 6, calls 6 and 8 (8: after the Morton cells became cubes and the radius 32): RT_CTX_OPT_TREE_BUILDER = 1 -- the shadow rays' own tree built on the device (PLOC) -- its tests, and against the host-built tree on configs 4, 2, 5:
# upload stages, steps per proxy ray (the tree report), shadow trace alone, the job.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_call08
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 600 python -m pytest tests/test_gpu_device_fold.py -q -m gpu -p no:cacheprovider -s > $O/pytest_device_fold.log 2>&1; el device fold + tree tests: $(grep -aE "passed|failed|rror" $O/pytest_device_fold.log | tail -1)
grep -aE "^E  |^FAILED|device-built / host-built" $O/pytest_device_fold.log | head -30
ARGS="--steps 4 --warmup 1 --no-cpu-baseline --per-frame-frames 0 --surface-area-fold-steps 2 --cold-job-spp 256"
run() { # name, config, extra args
  timeout 400 python bench.py --config $2 $ARGS $3 > $O/$1_cfg$2.json 2>> $O/bench.err
  el $1 cfg $2: $(python -c "
import json; d=json.loads(open('$O/$1_cfg$2.json').read().strip().splitlines()[-1]); k=d['roofline']['live_isolated']['kernel_ms_per_spp']; c=d['config']
print(d['value'], k, 'sa-fold', (d.get('surface_area_fold') or {}).get('value'), 'cold', {x: v for x, v in (d.get('cold_job') or {}).items() if x not in ('what', 'trees')}, c.get('setup_breakdown'), [t for t in c.get('trees', []) if t.startswith('upload') or t.startswith('shadow')], d['adaptation'].get('seconds_to_adapted'))" 2>&1 | tail -1)
}
for cfg in 4 2 5 3; do
  run host_1 $cfg "--tree-builder 0"; run device_1 $cfg "--tree-builder 1"
  run host_2 $cfg "--tree-builder 0"; run device_2 $cfg "--tree-builder 1"
done
run device_forced 5 "--tree-builder 1 --shadow-tree 2"
run host_forced 5 "--tree-builder 0 --shadow-tree 2"
el all done
