#!/bin/bash
# Round 4, call 27 (the round's last GPU minutes): RT_CTX_OPT_ADAPTIVE_FOLD on the device -- the bench line with the fold adapted to the frame's
# own rays (full-frame parity against oracle/_ref included), its GPU tests, the whole GPU suite (every sixth fuzz seed adapts), and the default
# bench line on the same box for the A/B.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_call27
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
show() { python -c "
import json; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], d['per_frame']['mrays_per_s'], d['parity']['bit_identical'], r['live_isolated']['kernel_ms_per_spp'], d['config']['trees'][-1][:260])"; }
timeout 150 python bench.py --adaptive-fold 3 --no-cpu-baseline > $O/bench_adaptive_fold.json 2> $O/bench_adaptive_fold.err; el bench adaptive: rc=$? $(show $O/bench_adaptive_fold.json 2>&1 | tail -1)
timeout 120 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k adaptive_fold -p no:cacheprovider > $O/pytest_adaptive_fold.log 2>&1; el adaptive tests: $(tail -1 $O/pytest_adaptive_fold.log)
timeout 300 python -m pytest tests -x -q -m gpu -k "not adaptive_fold_is_adopted" -p no:cacheprovider > $O/pytest_gpu_full.log 2>&1; el suite: $(tail -1 $O/pytest_gpu_full.log)
timeout 100 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; el bench default: rc=$? $(show $O/bench_default.json 2>&1 | tail -1)
tail -30 $O/pytest_adaptive_fold.log | cut -c1-300
tail -5 $O/bench_adaptive_fold.err | cut -c1-300
el all done
