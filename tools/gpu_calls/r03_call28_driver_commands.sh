#!/bin/bash
# Round 3, call 28: the driver's commands on the committed tree (suite incl. the new two-tile bench test, smoke, python bench.py
# with the committed counters file), the N = 2 bench line with its new `parity` leg (two ranks on the box's one GPU), and
# 6000 more fuzz seeds on the SAH-collapsed tree.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_call28
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror|FAILED" | tail -4 > $O/pytest_gpu.log; el suite: $(tail -1 $O/pytest_gpu.log)
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; el smoke: $(tail -1 $O/smoke.log)
( time python bench.py ) > $O/bench.json 2> $O/bench.err; el bench: $(python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], 'Mrays/s; roofline', {k: r.get(k) for k in ('bound','achieved','peak','frac','traffic','stale')}, 'ceiling', r['latency_ceiling']['frac_of_ceiling'], 'per_frame', d['per_frame']['mrays_per_s'], 'parity', d['parity']['bit_identical'], d['parity']['rel_l2_vs_libm_build'])" 2>&1 | tail -1)
grep real $O/bench.err
( time python bench.py --gpus 2 --debug-shared-gpu --steps 2 --samples-per-step 32 ) > $O/bench_2rank_shared_gpu.json 2> $O/bench2.err; el 2 ranks: $(python -c "
import json; d=json.loads(open('$O/bench_2rank_shared_gpu.json').read().strip().splitlines()[-1])
print(d['value'], 'Mrays/s n_gpus', d['n_gpus'], 'ranks', d['ranks']['render_ms'], d['ranks']['rows'], 'parity', d['parity'])" 2>&1 | tail -1)
grep real $O/bench2.err
( RT_FUZZ_SEEDS=9000 timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -n 32 -p no:cacheprovider -k "not (seed0- or seed1- or seed2-)" 2>&1 | grep -aE "passed|failed|rror|Timeout" | tail -3 ) > $O/fuzz_9000_seeds.log 2>&1; el fuzz: $(tail -1 $O/fuzz_9000_seeds.log)
el all done
