#!/bin/bash
# Round 5, call 6: where does the 2 % between two code objects with the SAME schedule (pre-quorum library vs the current one at quorum 1, call 5)
# come from?  The hot loop's placement: variants with loop C's header aligned to 64 / 256 bytes and with every block aligned; then the loop
# thresholds around 32 : 8 once more, now with the refill quorum in place.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_call06
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
line() { python - <<PY
import json
try:
    d = json.loads(open("$O/$1.json").read().strip().splitlines()[-1])
    k = (d["roofline"].get("live_isolated") or d["roofline"]["live"])["kernel_ms_per_spp"]
    print("$1: %.1f Mrays/s, alone %s" % (d["value"], k))
except Exception as e:
    print("$1: FAILED", e)
PY
}
Q="--steps 4 --no-cpu-baseline --per-frame-frames 0 --surface-area-fold-steps 0 --moving-camera-frames 0"
cp raytracing_amd/librt_hip.so $O/librt_hip_current.so
for v in current r05_pre_quorum r05_align6 r05_align8 r05_alignblocks current r05_pre_quorum r05_align6; do
  if [ $v = current ]; then cp $O/librt_hip_current.so raytracing_amd/librt_hip.so; else cp raytracing_amd/variants/$v/librt_hip.so raytracing_amd/librt_hip.so; fi
  n=$(ls $O | grep -c "^v_${v}_")
  timeout 300 python bench.py $Q > $O/v_${v}_$n.json 2>> $O/bench.err; el $(line v_${v}_$n)
done
cp $O/librt_hip_current.so raytracing_amd/librt_hip.so
for t in 0x0824 0x081C 0x0C20 0x0620; do
  timeout 300 python bench.py $Q --trace-tune $t > $O/t_$t.json 2>> $O/bench.err; el $(line t_$t)
done
for rq in 12 20; do
  timeout 300 python bench.py $Q --refill-quorum $rq > $O/rq_$rq.json 2>> $O/bench.err; el $(line rq_$rq)
done
rm -f $O/librt_hip_current.so
