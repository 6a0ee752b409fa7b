#!/bin/bash
# Round 4, call 18: same-box A/B of the headline: the library before the chunk-refill change against the current one.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_call18
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
v() { python -c "
import json; d=json.loads(open('$O/$1.json').read().strip().splitlines()[-1]); print(d['value'], (d['roofline'].get('live_isolated') or {}).get('kernel_ms_per_spp'))"; }
cp raytracing_amd/librt_hip.so /tmp/committed.so
for rep in 1 2; do for lib in committed before_chunk_refill; do
  if [ $lib = committed ]; then cp /tmp/committed.so raytracing_amd/librt_hip.so; else cp raytracing_amd/variants/$lib/librt_hip.so raytracing_amd/librt_hip.so; fi
  python bench.py --steps 3 --no-cpu-baseline --per-frame-frames 0 > $O/b_${lib}_$rep.json 2>> $O/bench.err; el $lib $rep: $(v b_${lib}_$rep)
done; done
cp /tmp/committed.so raytracing_amd/librt_hip.so
el all done
