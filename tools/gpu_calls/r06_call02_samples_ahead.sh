#!/bin/bash
# Round 6, call 2: RT_OPT_SAMPLES_AHEAD on the device for the first time -- its tests, the whole suite with it as HIPPathTraceIntegrator's default,
# the per-frame legs (off / automatic depth / fixed depths / one stream per bank) on configs 4, 5, 2, 3; the non-temporal queue accesses as
# build variants (RT_EXPERIMENT_NT) on configs 5 and 4; the sized read-request counters (the exact HBM read bytes) on the calibration kernels and
# on configs 4 and 5.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_call02
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 600 python -m pytest tests/test_gpu_samples_ahead.py tests/test_gpu_frame_kernel.py -x -q -m gpu -p no:cacheprovider > $O/pytest_ahead.log 2>&1; el samples-ahead + frame-kernel tests: $(grep -aE "passed|failed|rror" $O/pytest_ahead.log | tail -1)
grep -aE "^E |Error|assert" $O/pytest_ahead.log | head -12
timeout 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/pytest_gpu.log 2>&1; el suite: $(grep -aE "passed|failed|rror" $O/pytest_gpu.log | tail -1)
grep -aE "^E |^FAILED" $O/pytest_gpu.log | head -12
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; el smoke: $(tail -1 $O/smoke.log)
pf() { # config, samples-ahead value, frames
  timeout 400 python bench.py --config $1 --per-frame-only --per-frame-frames $3 --moving-camera-frames 0 --samples-ahead $2 > $O/pf_cfg$1_ahead$2.json 2>> $O/bench.err
  el cfg $1 ahead $2: $(python -c "
import json; d=json.loads(open('$O/pf_cfg$1_ahead$2.json').read().strip().splitlines()[-1])['per_frame']; a=d.get('samples_ahead') or {}
print(d['ms_per_frame'], 'ms/frame', d['mrays_per_s'], 'Mrays/s', 'k_frame frames', d['frames_through_k_frame'], 'replayed', a.get('frames_replayed_from_a_batch'), 'median/p99/max', a.get('ms_per_call_median'), a.get('ms_per_call_p99'), a.get('ms_per_call_max'), 'same bits', a.get('bit_identical_to_rt_integrate_of_the_same_samples'))" 2>&1 | tail -1)
}
for a in 0 1 8 16 264 272 32; do pf 4 $a 192; done
for a in 0 1 2 260 8; do pf 5 $a 64; done
for a in 0 1; do pf 2 $a 192; pf 3 $a 192; pf 1 $a 192; done
cp raytracing_amd/librt_hip.so $O/base_librt_hip.so
ARGS="--steps 4 --warmup 1 --no-cpu-baseline --per-frame-frames 0 --surface-area-fold-steps 0"
for rep in 1 2; do
for v in base nt1 nt2 nt3; do
  if [ $v = base ]; then cp $O/base_librt_hip.so raytracing_amd/librt_hip.so; else cp raytracing_amd/variants/$v/librt_hip.so raytracing_amd/librt_hip.so; fi
  for cfg in 5 4; do
    timeout 300 python bench.py --config $cfg $ARGS > $O/nt_${v}_cfg${cfg}_$rep.json 2>> $O/bench.err
    el $v cfg $cfg rep $rep: $(python -c "
import json; d=json.loads(open('$O/nt_${v}_cfg${cfg}_$rep.json').read().strip().splitlines()[-1]); k=d['roofline']['live_isolated']['kernel_ms_per_spp']; print(d['value'], k)" 2>&1 | tail -1)
  done
done
done
cp $O/base_librt_hip.so raytracing_amd/librt_hip.so; rm $O/base_librt_hip.so
( cd /tmp && export TMPDIR=/tmp
  timeout 120 $R/tools/bin/stream_mb 2 > $O/stream_mb.log 2>&1
  timeout 200 rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum --output-format csv -d $O/cal_fetch -o cal -- $R/tools/bin/stream_mb 2 > $O/cal_fetch.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ_sum --output-format csv -d $O/cal_write -o cal -- $R/tools/bin/stream_mb 2 > $O/cal_write.log 2>&1
  timeout 200 rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $O/cal_sized -o cal -- $R/tools/bin/stream_mb 2 > $O/cal_sized.log 2>&1
  for cfg in 4 5; do
    timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $O/rdreq_cfg$cfg -o rdreq -- python $R/bench.py --config $cfg --steps 2 --warmup 1 --overlap-shadow 0 --no-cpu-baseline --per-frame-frames 0 --surface-area-fold-steps 0 > $O/rdreq_cfg$cfg.log 2>&1
  done
)
python tools/stream_calibration.py $O $O/r06_fetch_size_calibration.json 2>&1 | tail -8
for cfg in 4 5; do echo "#### rdreq cfg $cfg"; python tools/pmc_summary.py $O/rdreq_cfg$cfg; done > $O/rdreq_summary.txt 2>&1
find $O -name "*.csv" -size +2M -delete
el all done
