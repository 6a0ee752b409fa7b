#!/bin/bash
# Round 4, call 15: can k_shade of one half-batch run beside the traces of the other?  Two pipes (RT_OPT_PIPELINES 2), fewer
# persistent trace waves per CU to leave wave slots, 512- and 256-thread shade blocks (variant library).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_call15
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
v() { python -c "
import json; d=json.loads(open('$O/$1.json').read().strip().splitlines()[-1]); print(d['value'], d['config']['pipelines'])"; }
cp raytracing_amd/librt_hip.so /tmp/committed.so
for lib in committed shade256; do
  if [ $lib = committed ]; then cp /tmp/committed.so raytracing_amd/librt_hip.so; else cp raytracing_amd/variants/$lib/librt_hip.so raytracing_amd/librt_hip.so; fi
  for p in 1 2; do for w in 0 22 18; do
    python bench.py --steps 3 --no-cpu-baseline --per-frame-frames 0 --pipelines $p --trace-waves $w > $O/b_${lib}_p${p}_w$w.json 2>> $O/bench.err; el $lib pipes $p trace-waves $w: $(v b_${lib}_p${p}_w$w)
  done; done
done
cp /tmp/committed.so raytracing_amd/librt_hip.so
grep -v amdgpu.ids $O/bench.err | tail -3
el all done
