#!/bin/bash
# round 2, GPU call 4: k_trace_w4 tuning + counters
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_call4
mkdir -p $O
cd $R
timeout 900 python tools/trace_variants.py --config 4 --slots 128 --spp 128 --variants 10 --tune 16:4,24:8,32:8,32:16,40:12,48:16 > $O/w4_tune.log 2>&1
cat $O/w4_tune.log
timeout 900 python tools/trace_variants.py --config 4 --slots 128 --spp 128 --variants 10 --tune 32:8 --waves 12,16,20,24 > $O/w4_waves.log 2>&1
cat $O/w4_waves.log
cd /tmp && export TMPDIR=/tmp
EXTRA="--config 4 --slots 128 --spp 128 --tune 32:8"
run() { name=$1; v=$2; shift 2; timeout 600 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_$name -o $name -- python $R/tools/trace_variants.py --variants $v $EXTRA > $O/pmc_$name.log 2>&1; }
run sq_v10 10 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE
run tcp_v10 10 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
run sq_v8 8 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE
run tcp_v8 8 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
for n in sq_v10 tcp_v10 sq_v8 tcp_v8; do python $R/tools/pmc_summary.py $O/pmc_$n; done > $O/pmc_summary.txt 2>&1
cat $O/pmc_summary.txt
find $O -name "*.csv" -size +2M -delete
