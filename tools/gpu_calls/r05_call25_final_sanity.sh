#!/bin/bash
# Round 5, call 25: the driver's commands once more on the final tree (same code object as call 20's evidence: the slow-walk experiment of calls 23 / 24 was reverted),
# and a longer fuzz campaign (3000 seeds) on it.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_call25
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
python -c "
from raytracing_amd import codeobj; import json
print('code object', codeobj.code_object_sha256()[:16], 'counters', json.load(open('profiles/r05_trace_counters.json'))['_code_object_sha256'][:16])" 2>&1 | tail -1
timeout 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/pytest_gpu.log 2>&1; el suite: $(grep -aE "passed|failed|rror" $O/pytest_gpu.log | tail -1)
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; el smoke: $(tail -1 $O/smoke.log)
( time python bench.py ) > $O/bench_driver_command.json 2> $O/bench.err; el bench: $(python -c "
import json; d=json.loads(open('$O/bench_driver_command.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], d['per_frame']['ms_per_frame'], d['parity']['bit_identical'], r['frac'], r['stale'], d['cpu_baseline']['value'])" 2>&1 | tail -1)
grep real $O/bench.err
RT_FUZZ_SEEDS=3000 timeout 600 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -n 32 -p no:cacheprovider > $O/fuzz_3000_seeds.log 2>&1; el fuzz: $(tail -1 $O/fuzz_3000_seeds.log)
