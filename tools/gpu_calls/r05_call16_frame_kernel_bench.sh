#!/bin/bash
# Round 5, call 16: k_frame's tests incl. the measured choice (RT_OPT_FRAME_KERNEL = 255), the kernel-variant tests that go through the refactored
# w4_trace_body / shade_entry, the driver's bench command with the frame-kernel legs, the per-frame legs of configs 2 / 3 / 5, and a same-box A/B of
# the batch path against the library before the refactoring (variants/r05_pre_quorum = commit 31b6ecd's kernels).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_call16
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 900 python -m pytest tests/test_gpu_frame_kernel.py -q -m gpu -p no:cacheprovider > $O/pytest_frame_kernel.log 2>&1; el frame kernel tests: $(tail -1 $O/pytest_frame_kernel.log); grep -E "^E " $O/pytest_frame_kernel.log | head -10
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -n 8 -k "samples_in_flight_and_kernel_variants or golden or stage" -p no:cacheprovider > $O/pytest_variants.log 2>&1; el variants: $(tail -1 $O/pytest_variants.log)
timeout 500 python bench.py > $O/bench_default.json 2>> $O/bench.err; python - <<PY
import json
d = json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
p = d["per_frame"]
print("bench default: %.1f Mrays/s; per frame %.3f ms; frame kernel %s" % (d["value"], p["ms_per_frame"], json.dumps(p.get("frame_kernel"))[:700]))
PY
el bench default
for cfg in 2 3 5; do
  timeout 400 python bench.py --config $cfg --no-cpu-baseline --steps 2 --surface-area-fold-steps 0 --moving-camera-frames 0 > $O/bench_cfg$cfg.json 2>> $O/bench.err; python - <<PY
import json
d = json.loads(open("$O/bench_cfg$cfg.json").read().strip().splitlines()[-1])
p = d["per_frame"]; k = p.get("frame_kernel") or {}
print("cfg $cfg: per frame %.3f ms; forced %s; measured choice %s" % (p["ms_per_frame"], k.get("forced"), k.get("measured_choice")))
PY
  el cfg $cfg
done
Q="--steps 4 --no-cpu-baseline --per-frame-frames 0 --surface-area-fold-steps 0 --moving-camera-frames 0"
cp raytracing_amd/librt_hip.so $O/librt_hip_current.so
for v in current r05_pre_quorum current r05_pre_quorum; do
  if [ $v = current ]; then cp $O/librt_hip_current.so raytracing_amd/librt_hip.so; else cp raytracing_amd/variants/$v/librt_hip.so raytracing_amd/librt_hip.so; fi
  n=$(ls $O | grep -c "^ab_${v}_")
  timeout 300 python bench.py $Q > $O/ab_${v}_$n.json 2>> $O/bench.err; python -c "
import json; d=json.loads(open('$O/ab_${v}_$n.json').read().strip().splitlines()[-1]); print('$v', d['value'], d['roofline']['live_isolated']['kernel_ms_per_spp'])"
done
cp $O/librt_hip_current.so raytracing_amd/librt_hip.so; rm -f $O/librt_hip_current.so
el all done
