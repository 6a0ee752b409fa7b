#!/bin/bash
# Round 3, call 30: the Bistro-scale OBJ from disk through bench.py again, on the final library (SAH-collapsed wide tree built by
# several host threads): parse, BVH, collapse, upload, render -- WITH the CPU leg this time (parity of the loaded asset's frame
# against the reference's own kernels fed by the reference's own loader path is what `data: "real"` will mean on the day).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_call30
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
( timeout 400 python tools/obj_ingest_timing.py --triangles 2800000 --dir /tmp/rt_obj_ingest > $O/obj_ingest_timing.log 2>&1 ); el ingest: $(tail -1 $O/obj_ingest_timing.log)
timeout 600 python bench.py --scene /tmp/rt_obj_ingest/city.obj --steps 2 --cpu-seconds 5 > $O/bench_scene_from_disk.json 2> $O/bench_scene_from_disk.err; el scene: $(python -c "
import json; d=json.loads(open('$O/bench_scene_from_disk.json').read().strip().splitlines()[-1]); k=d['roofline']['live_isolated']['kernel_ms_per_spp']
print(d['value'], 'Mrays/s', d['data'], d['config']['triangles'], 'tris scene_s', d['config']['scene_s'], 'setup_s', d['config']['setup_s'], 'per-frame', d['per_frame']['mrays_per_s'], k, 'stale', d['roofline'].get('stale'), 'parity', (d.get('parity') or {}).get('bit_identical'), (d.get('parity') or {}).get('rel_l2_vs_libm_build'))" 2>&1 | tail -1)
tail -3 $O/bench_scene_from_disk.err
el all done
