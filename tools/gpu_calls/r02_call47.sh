#!/bin/bash
# Round 2, call 47: the wide tree's collapse rule (largest-area-first instead of two BVH2 levels per record) with the general
# per-octant order table: GPU suite, fuzz campaign on the wide-tree kernel, A/B against the two-level fold (RT_WIDE_BVH=2 in the
# environment) on configs 4 and 5, then -- the new rule is this build's default -- the evidence set on it.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02c_final
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror|FAILED" | tail -6 > $O/pytest_gpu.log; el suite: $(tail -1 $O/pytest_gpu.log)
( time RT_FUZZ_VARIANT=10 RT_FUZZ_SEEDS=1200 timeout 400 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -n 32 -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror|Timeout" | tail -3 ) > $O/fuzz_w4_greedy_1200_seeds.log 2>&1; el fuzz: $(grep -a "passed\|failed" $O/fuzz_w4_greedy_1200_seeds.log | tail -1)
ab() { name=$1; shift
  timeout 300 python bench.py --no-cpu-baseline "$@" > $O/ab_$name.json 2> $O/ab_$name.err; python - <<PY
import json
try:
    d = json.loads(open("$O/ab_$name.json").read().strip().splitlines()[-1])
    k = (d["roofline"].get("live_isolated") or d["roofline"]["live"])["kernel_ms_per_spp"]
    print("ab $name: %.1f Mrays/s  %.4f ms/spp | alone: closest %.4f shadow %.4f shade %.4f" % (d["value"], d["ms_per_spp"], k["trace_closest"], k["trace_shadow"], k["shade"]))
except Exception as e:
    print("ab $name: FAILED", e)
PY
}
ab greedy | tee -a $O/ab.log
RT_WIDE_BVH=2 ab fold | tee -a $O/ab.log
ab greedy_again | tee -a $O/ab.log
ab cfg5_greedy --config 5 | tee -a $O/ab.log
RT_WIDE_BVH=2 ab cfg5_fold --config 5 | tee -a $O/ab.log
ab cfg2_greedy --config 2 | tee -a $O/ab.log
RT_WIDE_BVH=2 ab cfg2_fold --config 2 | tee -a $O/ab.log
el ab done
line() { python - <<PY
import json
try:
    d = json.loads(open("$O/$1.json").read().strip().splitlines()[-1])
    p, c = d.get("parity") or {}, d.get("cpu_baseline") or {}
    print("$1: %.1f Mrays/s %.4f ms/spp, in flight %s, parity bit_identical=%s rel_l2=%s, cpu %s Mrays/s on %s threads" % (
        d["value"], d["ms_per_spp"], d["config"]["samples_in_flight"], p.get("bit_identical"), p.get("rel_l2"), c.get("value"), c.get("cores")))
except Exception as e:
    print("$1: FAILED", e)
PY
}
( time python bench.py ) > $O/bench.json 2> $O/bench.err; el $(line bench)
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -o stats -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/stats_default.log 2>&1; find $O/stats_default -name "*.csv" -size +3M -delete )
D=$O/pmc; mkdir -p $D
ARGS="--steps 2 --warmup 1 --overlap-shadow 0 --no-cpu-baseline"
( cd /tmp && export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -o stats -- python $R/bench.py $ARGS > $D/stats.log 2>&1
  run() { name=$1; shift; timeout 600 rocprofv3 --pmc "$@" --output-format csv -d $D/$name -o $name -- python $R/bench.py $ARGS > $D/$name.log 2>&1; }
  run busy SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
  run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE
  run ta TA_TA_BUSY_sum TA_BUSY_max GRBM_GUI_ACTIVE
  run fetch FETCH_SIZE TCC_EA0_RDREQ_sum
  run write WRITE_SIZE TCC_EA0_WRREQ_sum
  run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE
  run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
)
for n in sq busy ta tcp tcc fetch write; do echo "#### $n"; python tools/pmc_summary.py $D/$n; done > $D/summary.txt 2>&1
find $D -name "*.csv" -size +3M -delete
el stats and counters done
python bench.py --config 3 > $O/bench_cfg3.json 2>> $O/bench.err; el $(line bench_cfg3)
python bench.py --config 2 > $O/bench_cfg2.json 2>> $O/bench.err; el $(line bench_cfg2)
python bench.py --config 5 --cpu-seconds 5 > $O/bench_cfg5.json 2>> $O/bench.err; el $(line bench_cfg5)
tail -3 $O/bench.err
el all done
