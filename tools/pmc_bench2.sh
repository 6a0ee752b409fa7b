#!/bin/bash
# Round-2 counter passes over the headline bench command (rocprofv3 --pmc, every group in its own run,
# no tracing domains) + the same counters over tools/issue_microbench.hip, whose kernels saturate one
# unit each: that is what calibrates "busy" (tools/make_counters_json.py).
# usage (on the GPU box): tools/pmc_bench2.sh <outdir under gpurun_out> [bench.py args]
OUT=$1; shift
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/$OUT
mkdir -p $D
cd /tmp && export TMPDIR=/tmp
ARGS="$* --no-cpu-baseline"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -o stats -- python $R/bench.py $ARGS > $D/stats.log 2>&1
run() { name=$1; shift; timeout 900 rocprofv3 --pmc "$@" --output-format csv -d $D/$name -o $name -- python $R/bench.py $ARGS > $D/$name.log 2>&1; }
mb() { name=$1; what=$2; shift 2; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $D/mb_$name -o mb_$name -- $R/tools/bin/issue_mb $what > $D/mb_$name.log 2>&1; }
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE
run busy SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
run ta TA_TA_BUSY_sum TA_BUSY_max GRBM_GUI_ACTIVE
run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run fetch FETCH_SIZE TCC_EA0_RDREQ_sum
run write WRITE_SIZE TCC_EA0_WRREQ_sum
run derived VALUBusy SALUBusy MemUnitBusy MemUnitStalled
mb busy valu SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU GRBM_GUI_ACTIVE
mb salu salu SQ_ACTIVE_INST_SCA SQ_INSTS_SALU GRBM_GUI_ACTIVE
mb ta l1 TA_TA_BUSY_sum TA_BUSY_max TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE
mb derived_valu valu VALUBusy SALUBusy
mb derived_l1 l1 MemUnitBusy MemUnitStalled
for n in sq busy ta tcp tcc fetch write derived; do echo "#### $n"; python $R/tools/pmc_summary.py $D/$n; done > $D/summary.txt 2>&1
for n in mb_busy mb_salu mb_ta mb_derived_valu mb_derived_l1; do echo "#### $n"; python $R/tools/pmc_summary.py $D/$n --all; done > $D/summary_mb.txt 2>&1
find $D -name "*.csv" -size +3M -delete
ls $D
