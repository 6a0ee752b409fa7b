#!/bin/bash
# Round 6, call 35: the fuzz with its newest choice (every thirteenth rt_integrate seed also renders the frame as 2 - 4 tiles), seeds 0 .. 7 999.
O=gpurun_out/r06_call35; mkdir -p $O
RT_FUZZ_SEEDS=8000 timeout 3000 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -n 32 -p no:cacheprovider > $O/fuzz_8000_seeds.log 2>&1; tail -1 $O/fuzz_8000_seeds.log; grep -a "^FAILED" $O/fuzz_8000_seeds.log | head
