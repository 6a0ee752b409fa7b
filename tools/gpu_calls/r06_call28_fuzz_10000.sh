#!/bin/bash
# Round 6, call 28: a long fuzz campaign on the final tree (10 000 seeds: every 7th through the stage API with samples ahead, every 6th adapting its folds, every 5th through k_frame).
O=gpurun_out/r06_call28; mkdir -p $O
RT_FUZZ_SEEDS=10000 timeout 3000 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -n 32 -p no:cacheprovider > $O/fuzz_10000_seeds.log 2>&1; tail -1 $O/fuzz_10000_seeds.log
