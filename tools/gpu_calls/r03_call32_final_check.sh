#!/bin/bash
# Round 3, call 32: the driver's commands on the final committed tree (suite, smoke, python bench.py).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_call32
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror|FAILED" | tail -4 > $O/pytest_gpu.log; el suite: $(tail -1 $O/pytest_gpu.log)
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; el smoke: $(tail -1 $O/smoke.log)
( time python bench.py ) > $O/bench.json 2> $O/bench.err; el bench: $(python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], 'Mrays/s; roofline', {k: r.get(k) for k in ('bound','achieved','peak','frac','traffic','stale')}, 'ceiling', r['latency_ceiling']['frac_of_ceiling'], 'per_frame', d['per_frame']['mrays_per_s'], 'parity', d['parity']['bit_identical'], d['parity']['rel_l2_vs_libm_build'], 'cpu', d['cpu_baseline']['value'])" 2>&1 | tail -1)
grep real $O/bench.err
el all done
