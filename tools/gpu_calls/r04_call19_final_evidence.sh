#!/bin/bash
# Round 4, call 19: call 13 again on the FINAL library (+ chunk mode with refilled lanes in the TAIL instance; own shadow tree, loop D, rt_frame_present, compact-log
# sub-pools): suite, kernel stats (one stream / default / per-frame), the counter passes -> profiles/r04_trace_counters.json (made
# on the box so that every line below reads counters of the code object it runs), the visit micro-benchmark, then the bench line of
# every config WITH the CPU leg, the bounded-state runs, the multi-rank plumbing and the C++ tiled path.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_final2
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
line() { python - <<PY
import json
try:
    d = json.loads(open("$O/$1.json").read().strip().splitlines()[-1])
    p, c, pf = d.get("parity") or {}, d.get("cpu_baseline") or {}, d.get("per_frame") or {}
    k = (d["roofline"].get("live_isolated") or d["roofline"]["live"])["kernel_ms_per_spp"]
    ce = d["roofline"].get("ceilings") or {}
    print("$1: %.1f Mrays/s %.4f ms/spp, in flight %s (%.1f GiB), per-frame %s Mrays/s (%s ms), parity bit_identical=%s rel_l2=%s vs libm %s, cpu %s Mrays/s on %s threads | alone: %s | ceilings %s binding %s frac %s | setup %s s | %s" % (
        d["value"], d["ms_per_spp"], d["config"]["samples_in_flight"], d["config"]["path_state_GB"], pf.get("mrays_per_s"), pf.get("ms_per_frame"), p.get("bit_identical"), p.get("rel_l2"),
        p.get("rel_l2_vs_libm_build"), c.get("value"), c.get("cores"), k, ce.get("grays"), ce.get("binding"), ce.get("frac_of_ceiling"), d["config"].get("setup_s"), d["config"].get("trees")))
except Exception as e:
    print("$1: FAILED", e)
PY
}
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror|FAILED" | tail -5 > $O/pytest_gpu.log; el suite: $(tail -1 $O/pytest_gpu.log)
( RT_FUZZ_SEEDS=2000 timeout 600 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -n 32 -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror|Timeout" | tail -3 ) > $O/fuzz_2000_seeds.log 2>&1; el fuzz: $(tail -1 $O/fuzz_2000_seeds.log)
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; el smoke: $(tail -1 $O/smoke.log)
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -o stats -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --per-frame-frames 0 > $O/stats_default.log 2>&1; find $O/stats_default -name "*.csv" -size +3M -delete )
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_per_frame -o stats -- python $R/bench.py --per-frame-only --per-frame-frames 16 > $O/stats_per_frame.log 2>&1; find $O/stats_per_frame -name "*.csv" -size +3M -delete )
D=$O/pmc; mkdir -p $D
ARGS="--steps 2 --warmup 1 --overlap-shadow 0 --no-cpu-baseline --per-frame-frames 0"
( cd /tmp && export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -o stats -- python $R/bench.py $ARGS > $D/stats.log 2>&1
  run() { name=$1; shift; timeout 600 rocprofv3 --pmc "$@" --output-format csv -d $D/$name -o $name -- python $R/bench.py $ARGS > $D/$name.log 2>&1; }
  run busy SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
  run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE
  run ta TA_TA_BUSY_sum TA_BUSY_max GRBM_GUI_ACTIVE
  run fetch FETCH_SIZE TCC_EA0_RDREQ_sum
  run write WRITE_SIZE TCC_EA0_WRREQ_sum
  run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE
  run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
)
for n in sq busy ta tcp tcc fetch write; do echo "#### $n"; python tools/pmc_summary.py $D/$n; done > $D/summary.txt 2>&1
python tools/make_counters_json.py $D 4 profiles/r04_trace_counters.json closest=0.453 shadow=0.479 shade=0.48 > $O/make_counters_json.log 2>&1; cp profiles/r04_trace_counters.json $O/r04_trace_counters.json; tail -3 $O/make_counters_json.log
find $D -name "*.csv" -size +3M -delete
timeout 300 tools/bin/visit_mb 0.93 0.87 4096 > $O/r04_visit_microbench.json 2> $O/visit_microbench.err; cp $O/r04_visit_microbench.json profiles/r04_visit_microbench.json
el stats, counters, micro-benchmark done
( time python bench.py ) > $O/bench.json 2> $O/bench.err; el $(line bench)
python bench.py --config 1 --steps 64 --warmup 4 > $O/bench_cfg1.json 2>> $O/bench.err; el $(line bench_cfg1)
python bench.py --config 2 > $O/bench_cfg2.json 2>> $O/bench.err; el $(line bench_cfg2)
python bench.py --config 3 > $O/bench_cfg3.json 2>> $O/bench.err; el $(line bench_cfg3)
python bench.py --config 4 --path-state-gb 32 --no-cpu-baseline > $O/bench_cfg4_32GiB.json 2>> $O/bench.err; el $(line bench_cfg4_32GiB)
python bench.py --config 4 --path-state-gb 16 --no-cpu-baseline > $O/bench_cfg4_16GiB.json 2>> $O/bench.err; el $(line bench_cfg4_16GiB)
python bench.py --config 4 --compact-log 1 --no-cpu-baseline > $O/bench_cfg4_compact_log.json 2>> $O/bench.err; el $(line bench_cfg4_compact_log)
python bench.py --config 4 --closest-tree 2 --cpu-seconds 6 > $O/bench_cfg4_tolerance_mode.json 2>> $O/bench.err; el $(line bench_cfg4_tolerance_mode)
python bench.py --gpus 2 --debug-shared-gpu --steps 2 --samples-per-step 32 --no-cpu-baseline > $O/bench_2rank_shared_gpu.json 2>> $O/bench.err; el 2rank: $(python -c "
import json; d=json.loads(open('$O/bench_2rank_shared_gpu.json').read().strip().splitlines()[-1]); print(d['value'], d['gather']['transport'][:40], d['ranks'].get('in_flight'), d['ranks'].get('rays_per_launch'), (d.get('parity') or {}).get('bit_identical'))")
raytracing_amd/rt_render -w 640 -h 360 --scene assets/CornellBox.obj --spp 64 --bounces 4 --gpus 1 --tiled 1 > $O/rt_render_tiled.log 2>&1; el rt_render: $(tail -1 $O/rt_render_tiled.log)
raytracing_amd/rt_render -w 640 -h 360 --scene assets/CornellBox.obj --spp 64 --bounces 4 --gpus 4 --shared_device 1 >> $O/rt_render_tiled.log 2>&1; el rt_render x4: $(tail -1 $O/rt_render_tiled.log)
python bench.py --config 5 --cpu-seconds 5 > $O/bench_cfg5.json 2>> $O/bench.err; el $(line bench_cfg5)
grep -v amdgpu.ids $O/bench.err | tail -3
el all done
