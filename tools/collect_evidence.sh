#!/bin/bash
# Collects the per-round evidence on the GPU box into gpurun_out/<name>/ :
#   usage: tools/collect_evidence.sh r01_final
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/pytest_gpu.log
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --config 2 > $O/bench_cfg2.json 2>> $O/bench.err
python bench.py --config 3 --no-cpu-baseline > $O/bench_cfg3.json 2>> $O/bench.err
python bench.py --config 5 --no-cpu-baseline > $O/bench_cfg5.json 2>> $O/bench.err
tools/pmc_bench.sh $1/pmc > $O/pmc_ls.txt 2>&1
python tools/pmc_summary.py $O/pmc > $O/pmc_summary.txt
python tools/tile_efficiency.py 2>&1 | grep "^tiles" > $O/tile_efficiency.log
cat $O/pytest_gpu.log; cut -c1-200 $O/bench.json
