#!/bin/bash
# Round 4, call 24: the suite with the two new GPU tests (mid-sample reads on a compact allocation, presented frames), and the tolerance
# mode with BOTH own-tree candidates for closest-hit rays (surface area / projected area along the lights, the cheaper by the proxy
# walk) against the reference's kernels on configs 4 / 2 / 3 / 5.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_call24
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
line() { python - <<PY
import json
try:
    d = json.loads(open("$O/$1.json").read().strip().splitlines()[-1])
    pf = d.get("per_frame") or {}
    k = (d["roofline"].get("live_isolated") or d["roofline"]["live"])["kernel_ms_per_spp"]
    par = d.get("parity") or {}
    print("$1: %.1f Mrays/s, per-frame %s | alone: %s | parity: %s | %s" % (d["value"], pf.get("mrays_per_s"), k,
        {x: par.get(x) for x in ("bit_identical", "differing_pixels", "rel_l2")} if par else None, d["config"].get("trees")))
except Exception as e:
    print("$1: FAILED", e)
PY
}
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror|FAILED" | tail -5 > $O/pytest_gpu.log; el suite: $(tail -1 $O/pytest_gpu.log)
python bench.py --steps 3 --per-frame-frames 48 --closest-tree 2 --cpu-seconds 6 > $O/bench_cfg4_tolerance.json 2>> $O/bench.err; el $(line bench_cfg4_tolerance)
for cfg in 2 3 5; do
  python bench.py --config $cfg --steps 2 --per-frame-frames 0 --closest-tree 2 --cpu-seconds 5 --libm-series '' > $O/bench_cfg${cfg}_tolerance.json 2>> $O/bench.err; el $(line bench_cfg${cfg}_tolerance)
done
grep -v amdgpu.ids $O/bench.err | tail -3
el all done
