#!/bin/bash
# Round 6, call 32: after the stage API's switch to the full log layout: the suite, then 10 000 fuzz seeds again (seed 5652 was the one failure of call 28).
O=gpurun_out/r06_call32; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; grep -aE "passed|failed" $O/pytest_gpu.log | tail -1
RT_FUZZ_SEEDS=10000 timeout 3000 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -n 32 -p no:cacheprovider > $O/fuzz_10000_seeds.log 2>&1; tail -1 $O/fuzz_10000_seeds.log; grep -a "^FAILED" $O/fuzz_10000_seeds.log | head
