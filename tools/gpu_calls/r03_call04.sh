#!/bin/bash
# round 3, GPU call 4: chunk mode of k_trace_w4 with STATIC chunk assignment (no hand-out atomics) on the per-frame path
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_call04
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
V1=4000000000
timeout 600 python tools/per_frame_sweep.py --config 4 --frames 32 --settings \
  v1_always:0:$V1:1:5 w4_rpl8:0x08000000:0:1:5 \
  chunk_full:0x00800000:0:1:5 chunk_rpl1:0x01800000:0:1:5 chunk_rpl2:0x02800000:0:1:5 chunk_rpl3:0x03800000:0:1:5 chunk_rpl4:0x04800000:0:1:5 chunk_rpl6:0x06800000:0:1:5 chunk_rpl8:0x08800000:0:1:5 chunk_rpl16:0x10800000:0:1:5 \
  chunk_rpl2_q1:0x02800101:0:1:5 chunk_rpl2_q16_4:0x02800410:0:1:5 chunk_rpl2_q48_16:0x02801030:0:1:5 chunk_rpl2_q64_8:0x02800840:0:1:5 chunk_rpl2_q32_1:0x02800120:0:1:5 chunk_rpl2_q32_32:0x02802020:0:1:5 \
  chunk_rpl2_no_overlap:0x02800000:0:0:5 chunk_rpl2_noresolve:0x02800000:0:1:5:0 \
  > $O/per_frame_sweep_cfg4.log 2>&1; el sweep4; cat $O/per_frame_sweep_cfg4.log
timeout 300 python tools/per_frame_sweep.py --config 2 --frames 32 --settings \
  v1_always:0:$V1:1:5 chunk_full:0x00800000:0:1:5 chunk_rpl1:0x01800000:0:1:5 chunk_rpl2:0x02800000:0:1:5 chunk_rpl4:0x04800000:0:1:5 chunk_rpl8:0x08800000:0:1:5 \
  > $O/per_frame_sweep_cfg2.log 2>&1; el sweep2; cat $O/per_frame_sweep_cfg2.log
timeout 300 python tools/per_frame_sweep.py --config 3 --frames 32 --settings \
  v1_always:0:$V1:1:5 chunk_rpl1:0x01800000:0:1:5 chunk_rpl2:0x02800000:0:1:5 chunk_rpl4:0x04800000:0:1:5 \
  > $O/per_frame_sweep_cfg3.log 2>&1; el sweep3; cat $O/per_frame_sweep_cfg3.log
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -p no:cacheprovider -k "variants_are_bit or axis_aligned or degenerate" 2>&1 | grep -aE "passed|failed|rror|FAILED|assert" | tail -5 > $O/pytest_subset.log; el subset: $(tail -1 $O/pytest_subset.log)
el all done
