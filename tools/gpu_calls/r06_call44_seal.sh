#!/bin/bash
# Round 6, call 44: the randomised tests once more at length on the final tree: 4000 API walks (three frames, tiles, AOVs, denoiser, adaptation under the walk) and
# fuzz seeds 40 000 .. 49 999 with every choice the fuzz has now (path-state bound, tiles).
O=gpurun_out/r06_call44; mkdir -p $O
RT_SEQ_SEEDS=4000 timeout 1500 python -m pytest tests/test_gpu_samples_ahead.py -k random_sequences -q -m gpu -n 16 -p no:cacheprovider > $O/api_walks_4000.log 2>&1; tail -1 $O/api_walks_4000.log; grep -a "^FAILED" $O/api_walks_4000.log | head -5
RT_FUZZ_FIRST=40000 RT_FUZZ_SEEDS=50000 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -n 32 -p no:cacheprovider > $O/fuzz_seeds_40000_49999.log 2>&1; tail -1 $O/fuzz_seeds_40000_49999.log; grep -a "^FAILED" $O/fuzz_seeds_40000_49999.log | head -5
