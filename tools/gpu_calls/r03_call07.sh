#!/bin/bash
# round 3, GPU call 7: the whole GPU suite on the current tree (new: full-frame 128-vs-8-in-flight, bench --scene, shared-device
# TiledRender), the driver's bench command with the CPU leg (parity + libm cross-pin), config 5 with the new automatic batch,
# bounded path state, the 2-rank plumbing run
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_call07
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 1200 python -m pytest tests -q -m gpu -x -p no:cacheprovider --durations=8 2>&1 | grep -aE "passed|failed|rror|FAILED|assert|s call|s setup" | tail -16 > $O/pytest_gpu.log; el suite: $(grep -aE "passed|failed" $O/pytest_gpu.log | tail -1); cat $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; el bench rc $?; python - <<PY
import json
try:
    d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print(d["value"], "Mrays/s", d["ms_per_spp"], "ms/spp in flight", d["config"]["samples_in_flight"], "GB", d["config"]["path_state_GB"])
    print("alone:", (r.get("live_isolated") or r["live"])["kernel_ms_per_spp"])
    print("roofline:", {k: r.get(k) for k in ("bound","achieved","peak","frac","traffic","stale")}, r.get("latency_ceiling"))
    print("per_frame:", d["per_frame"]["mrays_per_s"], d["per_frame"]["ms_per_frame"])
    print("parity:", d["parity"]); print("cpu:", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
except Exception as e:
    print("bench parse failed", e); print(open("$O/bench.err").read()[-1500:])
PY
b() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1]); k=(d["roofline"].get("live_isolated") or d["roofline"]["live"])["kernel_ms_per_spp"]
    print("$name: %.1f Mrays/s %.4f ms/spp in flight %d (%.1f GB) | alone: closest %.4f shadow %.4f shade %.4f | per-frame %s" % (d["value"], d["ms_per_spp"], d["config"]["samples_in_flight"], d["config"]["path_state_GB"], k["trace_closest"], k["trace_shadow"], k["shade"], (d.get("per_frame") or {}).get("mrays_per_s")))
except Exception as e:
    print("$name: FAILED", e); print(open("$O/bench_$name.err").read()[-800:])
PY
}
b cfg5_auto --config 5 --steps 3 --per-frame-frames 8; el cfg5
b cfg5_16 --config 5 --steps 3 --samples-in-flight 16 --per-frame-frames 0; el cfg5_16
b cfg4_32gb --config 4 --steps 4 --path-state-gb 32 --per-frame-frames 0; el 32gb
b cfg4_16gb --config 4 --steps 4 --path-state-gb 16 --per-frame-frames 0; el 16gb
b cfg2 --config 2 --steps 4; el cfg2
b cfg3 --config 3 --steps 4; el cfg3
timeout 600 python bench.py --gpus 2 --debug-shared-gpu --steps 2 --samples-per-step 16 --no-cpu-baseline > $O/bench_2rank_shared_gpu.json 2> $O/bench_2rank_shared_gpu.err; el 2rank rc $?; tail -c 1500 $O/bench_2rank_shared_gpu.json
el all done
