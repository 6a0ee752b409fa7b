#!/bin/bash
# Round 4, call 17: chunk mode with lanes refilled from the wave's own statically assigned chunks (RT_OPT_CHUNK_REFILL): suite + fuzz,
# the per-frame pattern on every config with it off / on, small batches, bounded path state, the headline.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_call17
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
pf() { python -c "
import json; d=json.loads(open('$O/$1.json').read().strip().splitlines()[-1]); print(d['per_frame']['mrays_per_s'], d['per_frame']['ms_per_frame'])"; }
v() { python -c "
import json; d=json.loads(open('$O/$1.json').read().strip().splitlines()[-1]); print(d['value'], (d['roofline'].get('live_isolated') or {}).get('kernel_ms_per_spp'))"; }
timeout 900 python -m pytest tests -q -m gpu -x -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror|FAILED" | tail -5 > $O/pytest_gpu.log; el suite: $(tail -1 $O/pytest_gpu.log)
( RT_FUZZ_SEEDS=2000 timeout 600 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -n 32 -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror|Timeout" | tail -3 ) > $O/fuzz_2000_seeds.log 2>&1; el fuzz: $(tail -1 $O/fuzz_2000_seeds.log)
for cfg in 4 2 3 5; do for r in 0 1; do
  python bench.py --config $cfg --steps 1 --no-cpu-baseline --per-frame-frames 64 --per-frame-only --chunk-refill $r > $O/pf_cfg${cfg}_r$r.json 2>> $O/bench.err; el per-frame cfg $cfg chunk-refill $r: $(pf pf_cfg${cfg}_r$r)
done; done
for r in 0 1; do
  python bench.py --steps 3 --no-cpu-baseline --per-frame-frames 0 --chunk-refill $r > $O/b_r$r.json 2>> $O/bench.err; el headline chunk-refill $r: $(v b_r$r)
  python bench.py --samples-in-flight 8 --steps 8 --samples-per-step 8 --no-cpu-baseline --per-frame-frames 0 --chunk-refill $r > $O/b8_r$r.json 2>> $O/bench.err; el 8 in flight chunk-refill $r: $(v b8_r$r)
  python bench.py --path-state-gb 16 --steps 3 --no-cpu-baseline --per-frame-frames 0 --chunk-refill $r > $O/b16g_r$r.json 2>> $O/bench.err; el 16 GiB chunk-refill $r: $(v b16g_r$r)
  python bench.py --config 5 --steps 2 --no-cpu-baseline --per-frame-frames 0 --chunk-refill $r > $O/b5_r$r.json 2>> $O/bench.err; el config 5 chunk-refill $r: $(v b5_r$r)
done
for t in 0 16 40 64; do
  python bench.py --steps 1 --no-cpu-baseline --per-frame-frames 64 --per-frame-only --chunk-refill 1 --tail-lanes $t > $O/pf_tail$t.json 2>> $O/bench.err; el per-frame chunk-refill 1 tail lanes $t: $(pf pf_tail$t)
done
grep -v amdgpu.ids $O/bench.err | tail -3
el all done
