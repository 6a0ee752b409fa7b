#!/bin/bash
# Round 4, call 4: same-box A/B of k_trace_w4 after the loop-D refactor (call 3 read 2 % slower than call 1 on another box):
# the committed kernel (step lambdas, amdgpu_waves_per_eu(7, 8): 72 VGPRs + 5 spilled dwords in phase A), the same with
# waves_per_eu(4, 8) (75 VGPRs: 6 waves per SIMD), and the kernel text of the commit before (no loop D).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_call04
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
line() { python - <<PY
import json
try:
    d = json.loads(open("$O/$1.json").read().strip().splitlines()[-1])
    pf = d.get("per_frame") or {}
    k = (d["roofline"].get("live_isolated") or d["roofline"]["live"])["kernel_ms_per_spp"]
    print("$1: %.1f Mrays/s %.4f ms/spp, per-frame %s Mrays/s %s ms | alone: %s" % (d["value"], d["ms_per_spp"], pf.get("mrays_per_s"), pf.get("ms_per_frame"), k))
except Exception as e:
    print("$1: FAILED", e)
PY
}
cp raytracing_amd/librt_hip.so /tmp/committed.so
for rep in 1 2; do
  for v in committed w4attr oldkernel; do
    if [ $v = committed ]; then cp /tmp/committed.so raytracing_amd/librt_hip.so; else cp raytracing_amd/variants/$v/librt_hip.so raytracing_amd/librt_hip.so; fi
    python bench.py --steps 3 --no-cpu-baseline --per-frame-frames 48 --tail-lanes 32 > $O/bench_cfg4_${v}_$rep.json 2>> $O/bench.err; el $(line bench_cfg4_${v}_$rep)
  done
done
cp /tmp/committed.so raytracing_amd/librt_hip.so
for t in 24 32 40 48; do
  python bench.py --steps 1 --no-cpu-baseline --per-frame-frames 96 --per-frame-only --tail-lanes $t > $O/pf_tail$t.json 2>> $O/bench.err; el tail $t: $(python -c "
import json; d=json.loads(open('$O/pf_tail$t.json').read().strip().splitlines()[-1]); print(d['per_frame']['mrays_per_s'], d['per_frame']['ms_per_frame'])")
done
for sl in 1500000 3000000 6000000 12000000; do
  python bench.py --steps 1 --no-cpu-baseline --per-frame-frames 96 --per-frame-only --tail-lanes 32 --small-launch-paths $sl > $O/pf_small$sl.json 2>> $O/bench.err; el small-launch $sl: $(python -c "
import json; d=json.loads(open('$O/pf_small$sl.json').read().strip().splitlines()[-1]); print(d['per_frame']['mrays_per_s'], d['per_frame']['ms_per_frame'])")
done
tail -3 $O/bench.err | grep -v amdgpu.ids
el all done
