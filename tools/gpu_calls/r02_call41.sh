#!/bin/bash
# Round 2, call 41: the GPU suite again with its summary line kept (call 40 cut it off), and A/B runs of compile-time
# variants of librt_hip.so (tools/build_variants.py): RAYMARGIN without the SDWA cell decode, k_shade with 256-thread
# blocks, k_shade touching its shading record before the partition barriers; stack-spill statistics.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_call41
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 600 python -m pytest tests -q -m gpu -x -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror" | tail -3 > $O/pytest_gpu_default.log; el default suite: $(tail -1 $O/pytest_gpu_default.log)
RT_TRACE_AUTO_WIDE_VARIANT=15 timeout 600 python -m pytest tests -q -m gpu -x -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror" | tail -3 > $O/pytest_gpu_auto15.log; el auto15 suite: $(tail -1 $O/pytest_gpu_auto15.log)
cp raytracing_amd/librt_hip.so /tmp/librt_hip_base.so
ab() { name=$1; lib=$2; shift 2
  if [ "$lib" = base ]; then cp /tmp/librt_hip_base.so raytracing_amd/librt_hip.so; else cp raytracing_amd/variants/$lib/librt_hip.so raytracing_amd/librt_hip.so; fi
  timeout 300 python bench.py --no-cpu-baseline "$@" > $O/ab_$name.json 2> $O/ab_$name.err; python - <<PY
import json
try:
    d = json.loads(open("$O/ab_$name.json").read().strip().splitlines()[-1])
    k = (d["roofline"].get("live_isolated") or d["roofline"]["live"])["kernel_ms_per_spp"]
    print("ab $name: %.1f Mrays/s  %.4f ms/spp | alone: closest %.4f shadow %.4f shade %.4f | spill lane-steps %d, rays to bvh2 %d, rays/step %.4g" % (
        d["value"], d["ms_per_spp"], k["trace_closest"], k["trace_shadow"], k["shade"], d["config"]["stack_spill_lane_steps"],
        d["config"]["rays_left_to_the_bvh2_kernel"], d["config"]["rays_per_step"]))
except Exception as e:
    print("ab $name: FAILED", e)
PY
}
ab base_v10_a base | tee -a $O/ab.log
ab base_v15 base --trace-variant 15 | tee -a $O/ab.log
ab nosdwa_v15 nosdwa --trace-variant 15 | tee -a $O/ab.log
ab blk256 blk256 | tee -a $O/ab.log
ab prefetch prefetch | tee -a $O/ab.log
ab base_v10_b base | tee -a $O/ab.log
ab spills_v10 base --steps 1 --warmup 0 --samples-per-step 16 | tee -a $O/ab.log
ab spills_v14 base --steps 1 --warmup 0 --samples-per-step 16 --trace-variant 14 | tee -a $O/ab.log
ab spills_v11 base --steps 1 --warmup 0 --samples-per-step 16 --trace-variant 11 | tee -a $O/ab.log
el ab done
cp /tmp/librt_hip_base.so raytracing_amd/librt_hip.so
( cd /tmp && export TMPDIR=/tmp
  for v in 10 15; do
    timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/sq_v$v -o sq -- python $R/bench.py --steps 1 --warmup 0 --overlap-shadow 0 --no-cpu-baseline --trace-variant $v > $O/sq_v$v.log 2>&1
    echo "#### variant $v"; python $R/tools/pmc_summary.py $O/sq_v$v | grep -a "k_trace_w4"
  done ) > $O/sq_summary.txt 2>&1
find $O -name "*.csv" -size +3M -delete
cat $O/sq_summary.txt
el all done
