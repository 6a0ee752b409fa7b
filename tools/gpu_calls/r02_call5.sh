#!/bin/bash
# round 2, GPU call 5: group tests, bench with parity, 2-rank shared-GPU run, counter passes
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_call5
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest_gpu.log
cat $O/pytest_gpu.log
( time timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err
tail -3 $O/bench.err; cut -c1-1500 $O/bench.json
timeout 900 python bench.py --gpus 2 --debug-shared-gpu --steps 2 --samples-per-step 32 --no-cpu-baseline > $O/bench_2rank_shared_gpu.json 2> $O/bench_2rank_shared_gpu.err
tail -3 $O/bench_2rank_shared_gpu.err; cut -c1-1200 $O/bench_2rank_shared_gpu.json
timeout 600 raytracing_amd/rt_render -w 640 -h 360 --scene assets/CornellBox.obj --spp 64 --bounces 4 --gpus 1 --tiled 1 > $O/rt_render_tiled.log 2>&1
cat $O/rt_render_tiled.log
tools/pmc_bench2.sh r02_call5/pmc --steps 2 --warmup 1 > $O/pmc_ls.txt 2>&1
cat $O/pmc/summary.txt
cat $O/pmc/summary_mb.txt
grep -i "k_trace\|k_shade\|k_flush\|k_raygen\|Name" $O/pmc/stats/*kernel_stats.csv | head -20
