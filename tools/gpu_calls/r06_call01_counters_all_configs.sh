#!/bin/bash
# Round 6, call 1 (the tree of round 5's end, code object 1624c2f8...): what VERDICT r05 found missing on the measurement side --
#  (1) FETCH_SIZE / WRITE_SIZE calibrated per access pattern against known byte counts (tools/stream_microbench.hip);
#  (2) the seven counter passes + kernel statistics for configs 5, 2 and 3 (config 4's are profiles/r05_trace_counters.json);
#  (3) rt_integrate at 2 / 4 / 8 / 16 samples in flight on configs 4 and 5: what a sample costs when k samples travel together
#      (the sizing of the sample-ahead mode of the frame-by-frame pattern).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_call01
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
( cd /tmp && export TMPDIR=/tmp
  timeout 120 $R/tools/bin/stream_mb 2 > $O/stream_mb.log 2>&1
  timeout 200 rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum --output-format csv -d $O/cal_fetch -o cal -- $R/tools/bin/stream_mb 2 > $O/cal_fetch.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ_sum --output-format csv -d $O/cal_write -o cal -- $R/tools/bin/stream_mb 2 > $O/cal_write.log 2>&1
  timeout 200 rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $O/cal_extra -o cal -- $R/tools/bin/stream_mb 2 > $O/cal_extra.log 2>&1
)
for n in cal_fetch cal_write cal_extra; do echo "#### $n"; python tools/pmc_summary.py $O/$n --all; done > $O/calibration_summary.txt 2>&1
el calibration: $(grep -c k_ $O/calibration_summary.txt) rows
cat $O/stream_mb.log | head -6
for cfg in 5 2 3; do
  D=$O/pmc_cfg$cfg; mkdir -p $D
  ARGS="--config $cfg --steps 2 --warmup 1 --overlap-shadow 0 --no-cpu-baseline --per-frame-frames 0 --surface-area-fold-steps 0"
  ( cd /tmp && export TMPDIR=/tmp
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -o stats -- python $R/bench.py $ARGS > $D/stats.log 2>&1
    run() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $D/$name -o $name -- python $R/bench.py $ARGS > $D/$name.log 2>&1; }
    run busy SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
    run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE
    run ta TA_TA_BUSY_sum TA_BUSY_max GRBM_GUI_ACTIVE
    run fetch FETCH_SIZE TCC_EA0_RDREQ_sum
    run write WRITE_SIZE TCC_EA0_WRREQ_sum
    run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE
    run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
    run wait SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
  )
  for n in sq busy ta tcp tcc fetch write wait; do echo "#### $n"; python tools/pmc_summary.py $D/$n; done > $O/pmc_summary_cfg$cfg.txt 2>&1
  cp $D/stats/stats_kernel_stats.csv $O/rocprofv3_kernel_stats_cfg$cfg.csv 2>/dev/null
  RT_COUNTERS_FOLD="adapted to the frame's rays" python tools/make_counters_json.py $D $cfg $O/r06_trace_counters.json closest=0.453 shadow=0.479 shade=0.48 > $O/make_counters_json_cfg$cfg.log 2>&1
  tail -3 $O/make_counters_json_cfg$cfg.log
  find $D -name "*.csv" -size +2M -delete
  el counters cfg $cfg
done
for k in 2 4 8 16; do
  timeout 300 python bench.py --samples-per-step $k --samples-in-flight $k --steps 16 --warmup 2 --no-cpu-baseline --per-frame-frames 0 --surface-area-fold-steps 0 > $O/sif_cfg4_$k.json 2>> $O/bench.err
  el cfg 4, $k in flight: $(python -c "
import json; d=json.loads(open('$O/sif_cfg4_$k.json').read().strip().splitlines()[-1]); print(d['value'], 'Mrays/s', round(d['ms_per_step']/$k, 3), 'ms per sample')" 2>&1 | tail -1)
done
for k in 1 2 4; do
  timeout 400 python bench.py --config 5 --samples-per-step $k --samples-in-flight $k --steps 8 --warmup 2 --no-cpu-baseline --per-frame-frames 0 --surface-area-fold-steps 0 > $O/sif_cfg5_$k.json 2>> $O/bench.err
  el cfg 5, $k in flight: $(python -c "
import json; d=json.loads(open('$O/sif_cfg5_$k.json').read().strip().splitlines()[-1]); print(d['value'], 'Mrays/s', round(d['ms_per_step']/$k, 3), 'ms per sample')" 2>&1 | tail -1)
done
el all done
