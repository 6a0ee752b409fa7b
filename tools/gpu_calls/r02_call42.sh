#!/bin/bash
# Round 2, call 42: the packet kernel (BVH2, node records through the scalar cache) on the coherent launches of the headline
# workload as it is now (128 samples of a pixel in one wave): closest bounce 0, shadow bounce 0, both, and bounces 0-1.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_call42
mkdir -p $O
cd $R
ab() { name=$1; shift
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
ab base | tee -a $O/ab.log
ab packet_closest0 --packet-bounces 0x001 | tee -a $O/ab.log
ab packet_shadow0 --packet-bounces 0x100 | tee -a $O/ab.log
ab packet_both0 --packet-bounces 0x101 | tee -a $O/ab.log
ab packet_both01 --packet-bounces 0x202 | tee -a $O/ab.log
ab cfg2_base --config 2 | tee -a $O/ab.log
ab cfg2_packet_both0 --config 2 --packet-bounces 0x101 | tee -a $O/ab.log
