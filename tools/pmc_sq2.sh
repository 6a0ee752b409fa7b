#!/bin/bash
# second SQ pass: where do the issue cycles go?
OUT=$1; VAR=$2; shift 2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$OUT
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/$OUT -o sq2 -- python $R/tools/trace_variants.py --variants $VAR "$@" > $R/gpurun_out/$OUT/sq2.log 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_VMEM SQ_IFETCH SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/$OUT -o sq3 -- python $R/tools/trace_variants.py --variants $VAR "$@" > $R/gpurun_out/$OUT/sq3.log 2>&1
