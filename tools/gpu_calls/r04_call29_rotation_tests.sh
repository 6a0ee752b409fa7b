#!/bin/bash
# Round 4, call 29: the round's last 18 GPU-seconds -- RT_CTX_OPT_ADAPTIVE_FOLD bit 3 (the shadow rays' tree rotated before it is folded) on the device:
# its three GPU tests (bit-identical radiance against the oracle, two views, a re-upload).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_call29
mkdir -p $O
cd $R
T="tests/test_gpu_parity.py::test_adaptive_fold_is_adopted_and_changes_no_bit"
timeout 13 python -m pytest "$T[15-1]" "$T[15-0]" "$T[9-2]" -x -q -p no:cacheprovider > $O/pytest_rotation.log 2>&1
tail -5 $O/pytest_rotation.log | cut -c1-400
