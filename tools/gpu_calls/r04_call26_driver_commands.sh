#!/bin/bash
# Round 4, call 26: the driver's own commands on the committed tree -- pytest -m gpu, smoke(), python bench.py -- and 6000 more fuzz seeds.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_call26
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror|FAILED" | tail -3 > $O/pytest_gpu.log; el suite: $(tail -1 $O/pytest_gpu.log)
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; el smoke: $(tail -1 $O/smoke.log)
( time python bench.py ) > $O/bench_driver_command.json 2> $O/bench.err; el bench: $(python -c "
import json; d=json.loads(open('$O/bench_driver_command.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], d['per_frame']['mrays_per_s'], d['parity']['bit_identical'], r['frac'], r['stale'], r['ceilings']['grays'], r['ceilings']['frac_of_ceiling'], r['live_isolated']['kernel_ms_per_spp'], d['cpu_baseline']['value'])")
grep real $O/bench.err
( RT_FUZZ_SEEDS=6000 timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -n 32 -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror|Timeout" | tail -3 ) > $O/fuzz_6000_seeds.log 2>&1; el fuzz: $(tail -1 $O/fuzz_6000_seeds.log)
el all done
