#!/bin/bash
# Round 6, call 12: hipMalloc vs the stream-ordered pool vs the virtual-memory calls for a 100 GiB buffer, one fresh process each (tools/alloc_modes_microbench.hip).
mkdir -p gpurun_out/r06_call12
for m in malloc async vmm vmm2m malloc; do timeout 300 tools/bin/alloc_modes $m 100 >> gpurun_out/r06_call12/alloc_modes.log 2>&1; done
timeout 300 tools/bin/alloc_modes malloc 24 >> gpurun_out/r06_call12/alloc_modes.log 2>&1
timeout 300 tools/bin/alloc_modes async 24 >> gpurun_out/r06_call12/alloc_modes.log 2>&1
cat gpurun_out/r06_call12/alloc_modes.log
