#!/bin/bash
# Round 2, call 44: what does k_shade pay for -- 2 more 16-byte accesses per surface hit (to the shading record's own line), 128 more
# vector instructions per surface hit, or fewer resident blocks (2 / 1 per CU instead of 3)?  (tools/build_variants.py)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_call44
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
ab shade_extra_access_2 shx_acc2 | tee -a $O/ab.log
ab shade_extra_valu_128 shx_valu128 | tee -a $O/ab.log
ab shade_2_blocks_per_cu sh_occ2 | tee -a $O/ab.log
ab shade_1_block_per_cu sh_occ1 | tee -a $O/ab.log
ab base_again base | tee -a $O/ab.log
