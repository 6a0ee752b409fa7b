#!/bin/bash
# round 3, GPU call 10: a Bistro-scale OBJ from disk through bench.py (opaque materials this time), a 12000-seed fuzz campaign
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_call10
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
( timeout 400 python tools/obj_ingest_timing.py --triangles 2800000 --dir /tmp/rt_obj_ingest > $O/obj_ingest_timing.log 2>&1 ); el ingest: $(tail -1 $O/obj_ingest_timing.log)
timeout 600 python bench.py --scene /tmp/rt_obj_ingest/city.obj --steps 2 --no-cpu-baseline > $O/bench_scene_from_disk.json 2> $O/bench_scene_from_disk.err; el scene: $(python -c "
import json; d=json.loads(open('$O/bench_scene_from_disk.json').read().strip().splitlines()[-1]); k=d['roofline']['live_isolated']['kernel_ms_per_spp']
print(d['value'], 'Mrays/s', d['data'], d['config']['triangles'], 'tris scene_s', d['config']['scene_s'], 'setup_s', d['config']['setup_s'], 'per-frame', d['per_frame']['mrays_per_s'], k, 'stale', d['roofline'].get('stale'))" 2>&1 | tail -1)
( RT_FUZZ_SEEDS=12000 timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -n 32 -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror|Timeout" | tail -3 ) > $O/fuzz_12000_seeds.log 2>&1; el fuzz: $(tail -1 $O/fuzz_12000_seeds.log)
el all done
