#!/bin/bash
# Round 4, call 21: the instance-dependent chunk-mode threshold (8 M rays in the loop-D instance) as the default: suite, per-frame,
# small batches, bounded path state, headline.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_call21
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
v() { python -c "
import json; d=json.loads(open('$O/$1.json').read().strip().splitlines()[-1]); pf = d.get('per_frame') or {}; print(d['value'], pf.get('mrays_per_s'))"; }
timeout 900 python -m pytest tests -q -m gpu -x -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror|FAILED" | tail -5 > $O/pytest_gpu.log; el suite: $(tail -1 $O/pytest_gpu.log)
python bench.py --steps 3 --no-cpu-baseline --per-frame-frames 64 > $O/b.json 2>> $O/bench.err; el headline + per-frame: $(v b)
for s in 2 4 8 16 32; do
  python bench.py --samples-in-flight $s --steps 8 --samples-per-step $s --no-cpu-baseline --per-frame-frames 0 > $O/b_$s.json 2>> $O/bench.err; el $s in flight: $(v b_$s)
done
for g in 32 16 8; do
  python bench.py --path-state-gb $g --steps 3 --no-cpu-baseline --per-frame-frames 0 > $O/b_${g}g.json 2>> $O/bench.err; el $g GiB: $(v b_${g}g)
done
python bench.py --config 5 --steps 2 --no-cpu-baseline --per-frame-frames 16 > $O/b5.json 2>> $O/bench.err; el config 5: $(v b5)
grep -v amdgpu.ids $O/bench.err | tail -3
el all done
