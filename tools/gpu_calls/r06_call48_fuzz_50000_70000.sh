#!/bin/bash
# Round 6, call 48: fuzz seeds 50 000 .. 69 999 with every choice the fuzz has (the round's last GPU minutes).
O=gpurun_out/r06_call48; mkdir -p $O
RT_FUZZ_FIRST=50000 RT_FUZZ_SEEDS=70000 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -n 32 -p no:cacheprovider > $O/fuzz_seeds_50000_69999.log 2>&1; tail -1 $O/fuzz_seeds_50000_69999.log; grep -a "^FAILED" $O/fuzz_seeds_50000_69999.log | head -5
