#!/bin/bash
# Collects the per-round evidence on the GPU box into gpurun_out/<name>/ :
#   usage: tools/collect_evidence.sh r02_final
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q 2>&1 | grep -a "passed\|failed\|error" | tail -3 > $O/pytest_gpu.log
( time python bench.py ) > $O/bench.json 2> $O/bench.err
python bench.py --config 2 > $O/bench_cfg2.json 2>> $O/bench.err
python bench.py --config 3 --no-cpu-baseline > $O/bench_cfg3.json 2>> $O/bench.err
python bench.py --config 5 --no-cpu-baseline > $O/bench_cfg5.json 2>> $O/bench.err
python bench.py --gpus 2 --debug-shared-gpu --steps 2 --samples-per-step 32 --no-cpu-baseline > $O/bench_2rank_shared_gpu.json 2>> $O/bench.err
raytracing_amd/rt_render -w 640 -h 360 --scene assets/CornellBox.obj --spp 64 --bounces 4 --gpus 1 --tiled 1 > $O/rt_render_tiled.log 2>&1
# counters with every launch on one stream (the profiler serialises kernels anyway; this keeps the attribution unambiguous)
tools/pmc_bench2.sh $1/pmc --steps 2 --warmup 1 --overlap-shadow 0 > $O/pmc_ls.txt 2>&1
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -o stats -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/stats_default.log 2>&1; find $O/stats_default -name "*.csv" -size +3M -delete )
python tools/tile_efficiency.py 2>&1 | grep "^tiles" > $O/tile_efficiency.log
python tools/tile_efficiency.py --steps 256 2>&1 | grep "^tiles" > $O/tile_efficiency_256spp.log
cat $O/pytest_gpu.log; for f in bench bench_cfg2 bench_cfg3 bench_cfg5 bench_2rank_shared_gpu; do python - <<PY
import json
try:
    d=json.loads(open("$O/$f.json").read().strip().splitlines()[-1])
    print("$f:", d["value"], "Mrays/s", d["ms_per_spp"], "ms/spp, in flight", d["config"]["samples_in_flight"], "parity", (d.get("parity") or {}).get("bit_identical"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print("$f: FAILED", e)
PY
done
cat $O/tile_efficiency.log $O/tile_efficiency_256spp.log
tail -3 $O/bench.err
