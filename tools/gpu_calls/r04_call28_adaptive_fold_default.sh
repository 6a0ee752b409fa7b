#!/bin/bash
# Round 4, call 28 (the round's last GPU minutes): RT_CTX_OPT_ADAPTIVE_FOLD is the default now (library: 1, bench.py: 3).  The whole GPU suite on
# that tree, the counter passes again on the adapted fold -> profiles/r04_trace_counters.json (tagged with the fold), then the driver's bench
# command reading them, then rocprofv3's kernel summary of the default run.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_call28
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 110 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/pytest_gpu.log 2>&1; el suite: $(grep -aE "passed|failed|rror" $O/pytest_gpu.log | tail -1)
D=$O/pmc; mkdir -p $D
ARGS="--steps 2 --warmup 1 --overlap-shadow 0 --no-cpu-baseline --per-frame-frames 0"
( cd /tmp && export TMPDIR=/tmp
  timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -o stats -- python $R/bench.py $ARGS > $D/stats.log 2>&1
  run() { name=$1; shift; timeout 60 rocprofv3 --pmc "$@" --output-format csv -d $D/$name -o $name -- python $R/bench.py $ARGS > $D/$name.log 2>&1; }
  run busy SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
  run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE
  run ta TA_TA_BUSY_sum TA_BUSY_max GRBM_GUI_ACTIVE
  run fetch FETCH_SIZE TCC_EA0_RDREQ_sum
  run write WRITE_SIZE TCC_EA0_WRREQ_sum
  run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE
  run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
)
RT_COUNTERS_FOLD="adapted to the frame's rays" python tools/make_counters_json.py $D 4 profiles/r04_trace_counters.json closest=0.453 shadow=0.479 shade=0.48 > $O/make_counters_json.log 2>&1
cp profiles/r04_trace_counters.json $O/r04_trace_counters.json; tail -3 $O/make_counters_json.log
for n in sq busy ta tcp tcc fetch write; do echo "#### $n"; python tools/pmc_summary.py $D/$n; done > $D/summary.txt 2>&1
cp $D/stats/stats_kernel_stats.csv $O/rocprofv3_kernel_stats_isolated.csv 2>/dev/null
find $D -name "*.csv" -size +2M -delete
el counters done
( time python bench.py ) > $O/bench_driver_command.json 2> $O/bench.err; el bench: $(python -c "
import json; d=json.loads(open('$O/bench_driver_command.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], d['per_frame']['mrays_per_s'], d['parity']['bit_identical'], r['frac'], r['stale'], r['ceilings']['grays'], r['ceilings']['frac_of_ceiling'], r['live_isolated']['kernel_ms_per_spp'], d['cpu_baseline']['value'], d['config']['trees'][-1][:200])" 2>&1 | tail -1)
grep real $O/bench.err
( cd /tmp && export TMPDIR=/tmp && timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -o stats -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --per-frame-frames 0 > $O/stats_default.log 2>&1; find $O/stats_default -name "*.csv" -size +2M -delete )
el all done
