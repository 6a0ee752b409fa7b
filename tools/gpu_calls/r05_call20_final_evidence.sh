#!/bin/bash
# Round 5, call 20: the final evidence again on the final code object (call 7 was before k_frame, the refactoring of the trace / shade bodies and the side-stream present):
# adaptive fold 25 / 27 by default, asynchronous probe, pointer-exchange adoption): the driver's commands (suite, smoke, bench), 2000 fuzz seeds,
# the counter passes -> profiles/r05_trace_counters.json, rocprofv3 kernel summaries, every config with its CPU leg, the per-frame timeline.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_call20
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/pytest_gpu.log 2>&1; el suite: $(grep -aE "passed|failed|rror" $O/pytest_gpu.log | tail -1)
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; el smoke: $(tail -1 $O/smoke.log)
RT_FUZZ_SEEDS=1500 timeout 600 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -n 32 -p no:cacheprovider > $O/fuzz_1500_seeds.log 2>&1; el fuzz: $(tail -1 $O/fuzz_1500_seeds.log)
D=$O/pmc; mkdir -p $D
ARGS="--steps 2 --warmup 1 --overlap-shadow 0 --no-cpu-baseline --per-frame-frames 0 --surface-area-fold-steps 0"
( cd /tmp && export TMPDIR=/tmp
  timeout 90 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -o stats -- python $R/bench.py $ARGS > $D/stats.log 2>&1
  run() { name=$1; shift; timeout 90 rocprofv3 --pmc "$@" --output-format csv -d $D/$name -o $name -- python $R/bench.py $ARGS > $D/$name.log 2>&1; }
  run busy SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
  run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE
  run ta TA_TA_BUSY_sum TA_BUSY_max GRBM_GUI_ACTIVE
  run fetch FETCH_SIZE TCC_EA0_RDREQ_sum
  run write WRITE_SIZE TCC_EA0_WRREQ_sum
  run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE
  run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
)
RT_COUNTERS_FOLD="adapted to the frame's rays" python tools/make_counters_json.py $D 4 profiles/r05_trace_counters.json closest=0.453 shadow=0.479 shade=0.48 > $O/make_counters_json.log 2>&1
cp profiles/r05_trace_counters.json $O/r05_trace_counters.json; tail -3 $O/make_counters_json.log
for n in sq busy ta tcp tcc fetch write; do echo "#### $n"; python tools/pmc_summary.py $D/$n; done > $O/pmc_summary.txt 2>&1
cp $D/stats/stats_kernel_stats.csv $O/rocprofv3_kernel_stats_isolated.csv 2>/dev/null
find $D -name "*.csv" -size +2M -delete
el counters done
( time python bench.py ) > $O/bench_driver_command.json 2> $O/bench.err; el bench: $(python -c "
import json; d=json.loads(open('$O/bench_driver_command.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], d['per_frame']['ms_per_frame'], d['per_frame']['frame_kernel']['default_went'], d['per_frame']['frame_kernel']['k_frame']['ms_per_frame'], d['per_frame']['moving_camera']['with_over_without'], d['parity']['bit_identical'], r['frac'], r['stale'], r['ceilings']['grays'], r['ceilings']['frac_of_ceiling'], r['live_isolated']['kernel_ms_per_spp'], d['cpu_baseline']['value'], d['surface_area_fold'], d['adaptation'])" 2>&1 | tail -1)
grep real $O/bench.err
( cd /tmp && export TMPDIR=/tmp && timeout 90 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -o stats -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --per-frame-frames 0 --surface-area-fold-steps 0 > $O/stats_default.log 2>&1; cp $O/stats_default/stats_kernel_stats.csv $O/rocprofv3_kernel_stats_overlap.csv 2>/dev/null; find $O/stats_default -name "*.csv" -size +2M -delete )
( cd /tmp && export TMPDIR=/tmp && timeout 90 rocprofv3 --kernel-trace --output-format csv -d $O/trace_pf -o pf -- python $R/bench.py --per-frame-only --per-frame-frames 6 --moving-camera-frames 0 > $O/trace_pf.log 2>&1
  f=$(find $O/trace_pf -name "*kernel_trace.csv" | head -1); n=$(python -c "
import csv; r=list(csv.DictReader(open('$f'))); print(len(r))"); python $R/tools/kernel_gantt.py $f $((n - 120)) 60 > $O/per_frame_gantt.log 2>&1; find $O/trace_pf -name "*.csv" -size +2M -delete )
el kernel summaries and the per-frame timeline
for cfg in 2 3 1 5; do
  extra=""; [ $cfg = 1 ] && extra="--steps 64 --warmup 4"; [ $cfg = 5 ] && extra="--cpu-seconds 5"
  timeout 500 python bench.py --config $cfg $extra > $O/bench_cfg$cfg.json 2>> $O/bench.err; el cfg $cfg: $(python -c "
import json; d=json.loads(open('$O/bench_cfg$cfg.json').read().strip().splitlines()[-1]); p=d['parity']; k=d['per_frame']['frame_kernel']; print(d['value'], d['per_frame']['ms_per_frame'], k['default_went'], k['stage_kernels']['ms_per_frame'], k['k_frame']['ms_per_frame'], k['k_frame']['bit_identical_to_the_default_leg'], p['bit_identical'], p.get('rel_l2_vs_libm_build'), p.get('reference_self_rel_l2'), p.get('median_pixel_rel_err_vs_libm_build'), d['surface_area_fold'])" 2>&1 | tail -1)
done
timeout 300 python bench.py --path-state-gb 32 --no-cpu-baseline --per-frame-frames 0 --surface-area-fold-steps 0 > $O/bench_cfg4_32GiB.json 2>> $O/bench.err; el 32 GiB: $(python -c "
import json; d=json.loads(open('$O/bench_cfg4_32GiB.json').read().strip().splitlines()[-1]); print(d['value'], d['config']['path_state_GB'], d['config']['samples_in_flight'])" 2>&1 | tail -1)
python - <<PY > $O/make_cache.log 2>&1
import argparse, bench
from raytracing_amd import host, scenes as S
for cfg in (4, 2):
    c = bench.CONFIGS[cfg]
    args = argparse.Namespace(config=cfg, scene=None, blob_tris=871_200, ball_tris=20_000, width=c["width"], height=c["height"], bounces=c["bounces"])
    raw = bench.build_scene(args, host, S, finish=False); raw.save_cache("/tmp/cfg%d.rtscene" % cfg); raw.close()
PY
timeout 300 raytracing_amd/rt_render -w 1920 -h 1080 --scene /tmp/cfg4.rtscene --bounces 8 --frames 192 2>&1 | tee $O/rt_render_frames_cfg4.log | tail -1
timeout 300 raytracing_amd/rt_render -w 1280 -h 720 --scene /tmp/cfg2.rtscene --bounces 8 --frames 192 2>&1 | tee $O/rt_render_frames_cfg2.log | tail -1
el all done
