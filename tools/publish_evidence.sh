#!/bin/bash
# Copies what tools/collect_evidence.sh left under gpurun_out/<name>/ into profiles/ (the tracked, judged copies) and
# regenerates profiles/r02_trace_counters.json.   usage: tools/publish_evidence.sh r02_final <f2 closest> <f2 shadow> <f2 shade>
set -e
N=$1; O=gpurun_out/$N
for f in bench bench_cfg2 bench_cfg3 bench_cfg5 bench_2rank_shared_gpu; do tail -1 $O/$f.json > profiles/${N}_$f.json; done
cp $O/pmc/summary.txt profiles/${N}_pmc_summary.txt
cp $O/pmc/summary_mb.txt profiles/${N}_pmc_summary_microbench.txt
cp $O/pytest_gpu.log profiles/${N}_pytest_gpu.log
cp $O/pmc/stats/stats_kernel_stats.csv profiles/${N}_rocprofv3_kernel_stats.csv
cp $O/stats_default/stats_kernel_stats.csv profiles/${N}_rocprofv3_kernel_stats_overlap.csv
cat $O/tile_efficiency.log > profiles/${N}_tile_efficiency.log
echo "--- 256 spp job" >> profiles/${N}_tile_efficiency.log
cat $O/tile_efficiency_256spp.log >> profiles/${N}_tile_efficiency.log
cp $O/rt_render_tiled.log profiles/${N}_rt_render_tiled.log
python tools/make_counters_json.py $O/pmc 4 profiles/r02_trace_counters.json closest=$2 shadow=$3 shade=$4
