#!/bin/bash
# rocprofv3 passes over the headline bench command: kernel trace + stats first, then the PMC
# counters in their own runs (no tracing domains), as MI355X_MICROARCH.md prescribes.
# usage (on the GPU box): tools/pmc_bench.sh <outdir under gpurun_out> [bench.py args]
OUT=$1; shift
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/$OUT
mkdir -p $D
cd /tmp && export TMPDIR=/tmp
ARGS="$*"
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o stats -- python $R/bench.py $ARGS --no-cpu-baseline > $D/stats.log 2>&1
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $D -o $name -- python $R/bench.py $ARGS --no-cpu-baseline > $D/$name.log 2>&1; }
run fetch FETCH_SIZE TCC_EA0_RDREQ_sum
run write WRITE_SIZE TCC_EA0_WRREQ_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE
run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
# calibration of FETCH_SIZE on this kernel's access pattern: random 64-byte records out of a
# 3.8 GB table (beyond the 256 MiB Infinity Cache), known byte count = records x 64
hipcc --offload-arch=gfx950 -O3 $R/tools/fetch_microbench.hip -o /tmp/fetch_mb 2> $D/calib_build.log
rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum --output-format csv -d $D -o calib -- /tmp/fetch_mb 60000001 > $D/calib.log 2>&1
ls $D
