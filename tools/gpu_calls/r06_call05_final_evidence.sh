#!/bin/bash
# Round 6, call 5: the final evidence on the final code object -- the driver's commands (suite, smoke, bench), 2000 fuzz seeds, the counter passes (now with the
# sized read requests: the exact HBM read bytes) for configs 4, 5, 2 and 3 -> profiles/r06_trace_counters.json, rocprofv3 kernel summaries, every config's
# bench line with its CPU leg, rt_render --frames, the N-way tile timing.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_call05
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 1200 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/pytest_gpu.log 2>&1; el suite: $(grep -aE "passed|failed|rror" $O/pytest_gpu.log | tail -1)
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; el smoke: $(tail -1 $O/smoke.log)
RT_FUZZ_SEEDS=2000 timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -n 32 -p no:cacheprovider > $O/fuzz_2000_seeds.log 2>&1; el fuzz: $(tail -1 $O/fuzz_2000_seeds.log)
for cfg in 4 5 2 3; do
  D=$O/pmc_cfg$cfg; mkdir -p $D
  ARGS="--config $cfg --steps 2 --warmup 1 --overlap-shadow 0 --no-cpu-baseline --per-frame-frames 0 --surface-area-fold-steps 0 --cold-job-spp 0"
  ( cd /tmp && export TMPDIR=/tmp
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -o stats -- python $R/bench.py $ARGS > $D/stats.log 2>&1
    run() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $D/$name -o $name -- python $R/bench.py $ARGS > $D/$name.log 2>&1; }
    run busy SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
    run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE
    run ta TA_TA_BUSY_sum TA_BUSY_max GRBM_GUI_ACTIVE
    run fetch FETCH_SIZE TCC_EA0_RDREQ_sum
    run write WRITE_SIZE TCC_EA0_WRREQ_sum
    run rdreq TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
    run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE
    run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
  )
  for n in sq busy ta tcp tcc fetch write rdreq; do echo "#### $n"; python tools/pmc_summary.py $D/$n; done > $O/pmc_summary_cfg$cfg.txt 2>&1
  cp $D/stats/stats_kernel_stats.csv $O/rocprofv3_kernel_stats_isolated_cfg$cfg.csv 2>/dev/null
  RT_COUNTERS_FOLD="adapted to the frame's rays" python tools/make_counters_json.py $D $cfg profiles/r06_trace_counters.json closest=0.453 shadow=0.479 shade=0.48 > $O/make_counters_json_cfg$cfg.log 2>&1
  tail -3 $O/make_counters_json_cfg$cfg.log
  find $D -name "*.csv" -size +2M -delete
  el counters cfg $cfg
done
cp profiles/r06_trace_counters.json $O/r06_trace_counters.json
( cd /tmp && export TMPDIR=/tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -o stats -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --per-frame-frames 0 --surface-area-fold-steps 0 --cold-job-spp 0 > $O/stats_default.log 2>&1; cp $O/stats_default/stats_kernel_stats.csv $O/rocprofv3_kernel_stats_overlap_cfg4.csv 2>/dev/null; find $O/stats_default -name "*.csv" -size +2M -delete )
( time python bench.py ) > $O/bench_driver_command.json 2> $O/bench.err; el bench: $(python -c "
import json; d=json.loads(open('$O/bench_driver_command.json').read().strip().splitlines()[-1]); p=d['per_frame']; a=p.get('samples_ahead') or {}; r=d['roofline']
print(d['value'], d['ms_per_step'], 'per frame', p['ms_per_frame'], p['mrays_per_s'], a.get('ms_per_call_median'), a.get('ms_per_call_p99'), a.get('bit_identical_to_rt_integrate_of_the_same_samples'), 'one sample per call', p['frame_kernel'].get('one_sample_per_call'), 'moving', p['moving_camera']['ms_per_frame'], p['moving_camera']['with_over_without'], 'parity', d['parity']['bit_identical'], 'roofline', r.get('frac'), r.get('achieved'), r.get('stale'), 'cold', {k: v for k, v in (d['cold_job'] or {}).items() if k not in ('what', 'trees')}, 'setup', d['config'].get('setup_breakdown'), 'adapt', d['adaptation'].get('seconds_to_adapted'), 'cpu', d['cpu_baseline']['value'])" 2>&1 | tail -1)
grep real $O/bench.err
for cfg in 2 3 1 5; do
  extra=""; [ $cfg = 1 ] && extra="--steps 64 --warmup 4"; [ $cfg = 5 ] && extra="--cpu-seconds 5"
  timeout 600 python bench.py --config $cfg $extra > $O/bench_cfg$cfg.json 2>> $O/bench.err; el cfg $cfg: $(python -c "
import json; d=json.loads(open('$O/bench_cfg$cfg.json').read().strip().splitlines()[-1]); p=d['parity']; f=d['per_frame']; a=f.get('samples_ahead') or {}; r=d['roofline']
print(d['value'], 'per frame', f['ms_per_frame'], a.get('bit_identical_to_rt_integrate_of_the_same_samples'), 'one per call', f['frame_kernel'].get('one_sample_per_call', {}).get('ms_per_frame'), f['frame_kernel'].get('default_went'), 'parity', p['bit_identical'], p.get('rel_l2_vs_libm_build'), 'roofline', r.get('frac'), r.get('stale'), 'sa fold', d['surface_area_fold'], 'cold', {k: v for k, v in (d['cold_job'] or {}).items() if k not in ('what', 'trees')})" 2>&1 | tail -1)
done
timeout 300 python bench.py --path-state-gb 32 --no-cpu-baseline --per-frame-frames 0 --surface-area-fold-steps 0 --cold-job-spp 0 > $O/bench_cfg4_32GiB.json 2>> $O/bench.err; el 32 GiB: $(python -c "
import json; d=json.loads(open('$O/bench_cfg4_32GiB.json').read().strip().splitlines()[-1]); print(d['value'], d['config']['path_state_GB'], d['config']['samples_in_flight'])" 2>&1 | tail -1)
python - <<PY > $O/make_cache.log 2>&1
import argparse, bench
from raytracing_amd import host, scenes as S
for cfg in (4, 2):
    c = bench.CONFIGS[cfg]
    args = argparse.Namespace(config=cfg, scene=None, blob_tris=871_200, ball_tris=20_000, width=c["width"], height=c["height"], bounces=c["bounces"])
    raw = bench.build_scene(args, host, S, finish=False); raw.save_cache("/tmp/cfg%d.rtscene" % cfg); raw.close()
PY
for a in 1 0; do timeout 300 raytracing_amd/rt_render -w 1920 -h 1080 --scene /tmp/cfg4.rtscene --bounces 8 --frames 192 --samples_ahead $a 2>&1 | tee $O/rt_render_frames_cfg4_ahead$a.log | tail -1; done
timeout 300 raytracing_amd/rt_render -w 1280 -h 720 --scene /tmp/cfg2.rtscene --bounces 8 --frames 192 2>&1 | tee $O/rt_render_frames_cfg2.log | tail -1
timeout 900 python tools/tile_efficiency.py --json $O/r06_tile_efficiency.json > $O/tile_efficiency.log 2>&1; el tile timing: $(tail -2 $O/tile_efficiency.log | tr '\n' ' ')
timeout 400 python tools/tile_efficiency.py --tiles 1,8 --spp 256 --pipelines 2 > $O/tile_efficiency_pipelines2.log 2>&1; el tile timing, 2 pipes: $(tail -1 $O/tile_efficiency_pipelines2.log)
el all done
