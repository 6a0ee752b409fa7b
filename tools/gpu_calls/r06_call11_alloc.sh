#!/bin/bash
# Round 6, call 11: what hipMalloc of the per-path buffers costs and when (tools/alloc_microbench.hip) -- the cold_job leg of call 10 showed 0.2 - 4.3 s.
mkdir -p gpurun_out/r06_call11
timeout 600 tools/bin/alloc_mb 100 > gpurun_out/r06_call11/alloc_mb_100.log 2>&1
cat gpurun_out/r06_call11/alloc_mb_100.log
