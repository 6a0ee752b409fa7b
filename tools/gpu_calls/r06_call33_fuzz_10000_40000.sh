#!/bin/bash
# Round 6, call 33: fuzz seeds 10 000 .. 39 999 (the first 10 000: call 32).
O=gpurun_out/r06_call33; mkdir -p $O
RT_FUZZ_FIRST=10000 RT_FUZZ_SEEDS=40000 timeout 3400 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -n 32 -p no:cacheprovider > $O/fuzz_seeds_10000_39999.log 2>&1; tail -1 $O/fuzz_seeds_10000_39999.log; grep -a "^FAILED" $O/fuzz_seeds_10000_39999.log | head
