#!/bin/bash
# round 3, GPU call 3: the per-frame path (one Integrate() per frame): chunk mode of k_trace_w4, page-locked resolve buffer,
# what the per-frame resolve costs
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_call03
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -p no:cacheprovider -k "variants_are_bit or axis_aligned or degenerate" 2>&1 | grep -aE "passed|failed|rror|FAILED|assert" | tail -5 > $O/pytest_subset.log; el subset: $(tail -1 $O/pytest_subset.log)
V1=4000000000
timeout 600 python tools/per_frame_sweep.py --config 4 --frames 32 --settings \
  v1_always:0:$V1:1:5 v1_noresolve:0:$V1:1:5:0 v1_no_overlap:0:$V1:0:5 \
  w4_rpl8:0x08000000:0:1:5 w4_rpl8_noresolve:0x08000000:0:1:5:0 \
  w4_chunk:0x00800000:0:1:5 w4_chunk_noresolve:0x00800000:0:1:5:0 w4_chunk_q1:0x00800101:0:1:5 w4_chunk_q16_4:0x00800410:0:1:5 w4_chunk_q48_16:0x00801030:0:1:5 w4_chunk_q64_1:0x00800140:0:1:5 \
  w4_chunk_rpl1:0x01800000:0:1:5 w4_chunk_rpl2:0x02800000:0:1:5 w4_chunk_rpl4:0x04800000:0:1:5 \
  > $O/per_frame_sweep_cfg4.log 2>&1; el sweep4; cat $O/per_frame_sweep_cfg4.log
timeout 300 python tools/per_frame_sweep.py --config 2 --frames 32 --settings \
  v1_always:0:$V1:1:5 v1_noresolve:0:$V1:1:5:0 w4_rpl8:0x08000000:0:1:5 w4_chunk:0x00800000:0:1:5 w4_chunk_q1:0x00800101:0:1:5 w4_chunk_rpl2:0x02800000:0:1:5 \
  > $O/per_frame_sweep_cfg2.log 2>&1; el sweep2; cat $O/per_frame_sweep_cfg2.log
timeout 300 python tools/per_frame_sweep.py --config 3 --frames 32 --settings \
  v1_always:0:$V1:1:5 w4_rpl8:0x08000000:0:1:5 w4_chunk:0x00800000:0:1:5 \
  > $O/per_frame_sweep_cfg3.log 2>&1; el sweep3; cat $O/per_frame_sweep_cfg3.log
el all done
