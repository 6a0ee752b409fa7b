#!/bin/bash
# Round 2, call 43: what does k_trace_w4 pay for -- one more 16-byte L1 access per node visit (to a line the visit fetches
# anyway) or 16 / 32 more vector instructions per node visit?  (tools/build_variants.py, RT_W4_EXTRA_ACCESS / RT_W4_EXTRA_VALU)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_call43
mkdir -p $O
cd $R
cp raytracing_amd/librt_hip.so /tmp/librt_hip_base.so
ab() { name=$1; lib=$2; shift 2
  if [ "$lib" = base ]; then cp /tmp/librt_hip_base.so raytracing_amd/librt_hip.so; else cp raytracing_amd/variants/$lib/librt_hip.so raytracing_amd/librt_hip.so; fi
  timeout 300 python bench.py --no-cpu-baseline "$@" > $O/ab_$name.json 2> $O/ab_$name.err; python - <<PY
import json
try:
    d = json.loads(open("$O/ab_$name.json").read().strip().splitlines()[-1])
    k = (d["roofline"].get("live_isolated") or d["roofline"]["live"])["kernel_ms_per_spp"]
    print("ab $name: %.1f Mrays/s  %.4f ms/spp | alone: closest %.4f shadow %.4f shade %.4f" % (d["value"], d["ms_per_spp"], k["trace_closest"], k["trace_shadow"], k["shade"]))
except Exception as e:
    print("ab $name: FAILED", e)
PY
}
ab base base | tee -a $O/ab.log
ab extra_access_1 xa1 | tee -a $O/ab.log
ab extra_access_2 xa2 | tee -a $O/ab.log
ab extra_valu_16 xv16 | tee -a $O/ab.log
ab extra_valu_32 xv32 | tee -a $O/ab.log
ab base_again base | tee -a $O/ab.log
