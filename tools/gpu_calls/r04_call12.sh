#!/bin/bash
# Round 4, call 12: the suite with the present test, a 4000-seed fuzz campaign whose variants now include the shadow tree's modes
# (measured / own forced / surface-area metric / shared) and loop D's instance and threshold, and the driver's bench command.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_call12
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror|FAILED" | tail -5 > $O/pytest_gpu.log; el suite: $(tail -1 $O/pytest_gpu.log)
( RT_FUZZ_SEEDS=4000 timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -n 32 -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror|Timeout" | tail -3 ) > $O/fuzz_4000_seeds.log 2>&1; el fuzz: $(tail -1 $O/fuzz_4000_seeds.log)
( time python bench.py ) > $O/bench.json 2> $O/bench.err; el bench: $(python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['per_frame']['mrays_per_s'], d['parity']['bit_identical'], d['roofline'].get('ceilings'), d['cpu_baseline']['value'])")
grep real $O/bench.err
el all done
