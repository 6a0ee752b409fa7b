#!/bin/bash
# Round 6, call 34: the fuzz with its new choice (every eleventh rt_integrate seed bounds the path state: chunked tiles, automatic compact log), seeds 0 .. 11 999.
O=gpurun_out/r06_call34; mkdir -p $O
RT_FUZZ_SEEDS=12000 timeout 3000 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -n 32 -p no:cacheprovider > $O/fuzz_12000_seeds.log 2>&1; tail -1 $O/fuzz_12000_seeds.log; grep -a "^FAILED" $O/fuzz_12000_seeds.log | head
