#!/bin/bash
# PMC passes over the trace kernels (counters in their own runs, no tracing domains).
# usage: tools/pmc_passes.sh <outdir> <variants> [extra args for trace_variants.py]
OUT=$1; VAR=$2; shift 2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/$OUT -o $name -- python $R/tools/trace_variants.py --variants $VAR $EXTRA > $R/gpurun_out/$OUT/$name.log 2>&1; }
mkdir -p $R/gpurun_out/$OUT
EXTRA="$*"
run fetch FETCH_SIZE TCC_HIT_sum
run tcc TCC_MISS_sum TCC_REQ_sum TCC_READ_sum TCC_EA0_RDREQ_sum
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE
run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
ls $R/gpurun_out/$OUT
