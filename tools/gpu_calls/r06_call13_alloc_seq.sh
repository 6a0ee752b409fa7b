#!/bin/bash
# Round 6, call 13: the bench's allocation sequence (large, half, large, quarter ...) in one process, hipMalloc vs the stream-ordered pool, fresh processes alternating.
mkdir -p gpurun_out/r06_call13
for m in seq-malloc seq-async seq-malloc seq-async; do timeout 300 tools/bin/alloc_modes $m 100 >> gpurun_out/r06_call13/alloc_seq.log 2>&1; echo >> gpurun_out/r06_call13/alloc_seq.log; done
timeout 300 tools/bin/alloc_mb 100 >> gpurun_out/r06_call13/alloc_seq.log 2>&1
cat gpurun_out/r06_call13/alloc_seq.log
