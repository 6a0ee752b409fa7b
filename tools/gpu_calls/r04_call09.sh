#!/bin/bash
# Round 4, call 9: how many 6 KiB one-wave blocks are really resident per CU (the visit micro-benchmark collapses from 24 to 26)?
# occupancy query + a 25-wave point; per-frame and headline sweeps of RT_OPT_TRACE_WAVES_PER_CU around it.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_call09
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 300 tools/bin/visit_mb 0.93 0.87 4096 > $O/visit_microbench.json 2> $O/visit_microbench.err; el visit_mb: $(python -c "
import json; d=json.load(open('$O/visit_microbench.json')); print('resident', d.get('resident_blocks_per_cu'), [(r['kernel'][:3], r['waves_per_cu'], round(r['gvisits_per_s'],1)) for r in d['runs']])")
for w in 26 25 24 22 20 16; do
  python bench.py --steps 1 --no-cpu-baseline --per-frame-frames 96 --per-frame-only --trace-waves $w > $O/pf_waves$w.json 2>> $O/bench.err; el per-frame waves $w: $(python -c "
import json; d=json.loads(open('$O/pf_waves$w.json').read().strip().splitlines()[-1]); print(d['per_frame']['mrays_per_s'], d['per_frame']['ms_per_frame'])")
done
for w in 26 25 24; do
  python bench.py --steps 3 --no-cpu-baseline --per-frame-frames 0 --trace-waves $w > $O/bench_waves$w.json 2>> $O/bench.err; el headline waves $w: $(python -c "
import json; d=json.loads(open('$O/bench_waves$w.json').read().strip().splitlines()[-1]); print(d['value'], (d['roofline'].get('live_isolated') or {}).get('kernel_ms_per_spp'))")
done
tail -3 $O/bench.err | grep -v amdgpu.ids
el all done
