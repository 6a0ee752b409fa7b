#!/bin/bash
# round 2, GPU call 1: parity of the new node layout + k_trace2, issue-rate calibration, variant sweep
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_call1
mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest_gpu.log
cat $O/pytest_gpu.log
timeout 300 tools/bin/issue_mb all > $O/issue_microbench.log 2>&1
cat $O/issue_microbench.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mb_sq -o mb -- $R/tools/bin/issue_mb valu > $O/pmc_mb_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mb_salu -o mb -- $R/tools/bin/issue_mb salu > $O/pmc_mb_salu.log 2>&1
timeout 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mb_tcp -o mb -- $R/tools/bin/issue_mb l1 > $O/pmc_mb_tcp.log 2>&1
cd $R
timeout 900 python tools/trace_variants.py --config 4 --slots 128 --spp 128 --variants 5,8,9 --tune 24:8,32:8,40:12,48:16,56:16,40:4,40:24,32:16 > $O/variants_cfg4.log 2>&1
cat $O/variants_cfg4.log
cd /tmp
EXTRA="--config 4 --slots 128 --spp 128"
run() { name=$1; v=$2; shift 2; timeout 600 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_$name -o $name -- python $R/tools/trace_variants.py --variants $v $EXTRA > $O/pmc_$name.log 2>&1; }
run sq_v8 8 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE
run tcp_v8 8 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
python $R/tools/pmc_summary.py $O/pmc_sq_v8 > $O/pmc_v8_summary.txt 2>&1
python $R/tools/pmc_summary.py $O/pmc_tcp_v8 >> $O/pmc_v8_summary.txt 2>&1
python $R/tools/pmc_summary.py $O/pmc_mb_sq --all > $O/pmc_mb_summary.txt 2>&1
python $R/tools/pmc_summary.py $O/pmc_mb_salu --all >> $O/pmc_mb_summary.txt 2>&1
python $R/tools/pmc_summary.py $O/pmc_mb_tcp --all >> $O/pmc_mb_summary.txt 2>&1
cat $O/pmc_v8_summary.txt $O/pmc_mb_summary.txt
# keep the pulled directory small
find $O -name "*.csv" -size +2M -delete
ls -la $O
