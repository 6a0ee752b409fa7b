#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_call29
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_headline_parity.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error|Timeout" | tail -5
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error|Timeout" | tail -5
for t in 0x3F000820 0x18000820 0x20000820; do
for s in 4 16 128; do
k=$((256 / s)); if [ $k -lt 2 ]; then k=2; fi
timeout 300 python bench.py --steps $k --warmup 1 --samples-per-step $s --samples-in-flight $s --trace-tune $t --no-cpu-baseline > $O/b.json 2> $O/b.err
python - <<PY
import json
try:
    d=json.loads(open("$O/b.json").read().strip().splitlines()[-1])
    print("tune $t samples per step $s:", d["value"], "Mrays/s", d["ms_per_spp"], "ms/spp", d["roofline"]["live"]["kernel_ms_per_spp"])
except Exception as e:
    print("tune $t samples per step $s: FAILED", e)
PY
done; done > $O/tail.log 2>&1
cat $O/tail.log
timeout 300 python tools/launch_timeline.py --in-flight 4,128 2>&1 | grep -E "samples in flight|all bounces|bounce 1:|bounce 7:|waves gone"
