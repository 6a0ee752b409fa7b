#!/bin/bash
# Round 5, call 10: is the presented image's copy a shader blit or an SDMA transfer -- in rt_render (C++) and in bench.py's process?
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_call10
mkdir -p $O
cd $R
python - <<PY > $O/make_cache.log 2>&1
import argparse, bench
from raytracing_amd import host, scenes as S
c = bench.CONFIGS[4]
args = argparse.Namespace(config=4, scene=None, blob_tris=871_200, ball_tris=20_000, width=c["width"], height=c["height"], bounces=c["bounces"])
raw = bench.build_scene(args, host, S, finish=False); raw.save_cache("/tmp/cfg4.rtscene"); raw.close()
PY
( cd /tmp && export TMPDIR=/tmp
  timeout 200 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/t_cpp -o t -- $R/raytracing_amd/rt_render -w 1920 -h 1080 --scene /tmp/cfg4.rtscene --bounces 8 --frames 24 > $O/t_cpp.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/t_py -o t -- python $R/bench.py --per-frame-only --per-frame-frames 24 --moving-camera-frames 0 > $O/t_py.log 2>&1 )
for w in cpp py; do echo "=== $w"; tail -2 $O/t_$w.log | cut -c1-200; grep -h "copyBuffer\|fillBuffer" $O/t_$w/*kernel_stats.csv | cut -c1-120; cat $O/t_$w/*memory_copy_stats.csv | cut -c1-160; f=$(ls $O/t_$w/*kernel_trace.csv); n=$(python -c "
import csv; print(len(list(csv.DictReader(open('$f')))))"); python tools/kernel_gantt.py $f $((n - 75)) 14; done
find $O -name "*.csv" -size +1M -delete
