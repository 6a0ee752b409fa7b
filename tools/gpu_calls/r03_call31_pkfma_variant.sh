#!/bin/bash
# Round 3, call 31: A/B of a compile-time variant of the library (built from a scratch copy of csrc/ into
# raytracing_amd/variants/pkfma/): the 24 slab-distance fmas of a closest-hit node visit as 12 v_pk_fma_f32 (two slots per
# instruction: 1.94 against 2 x 1.35 issue units, profiles/r03_call26_*; -12 vector instructions per visit, same registers).
# Parity tests + fuzz with the variant, then bench base / variant / base on one box; and three loop-threshold settings on the
# SAH tree (runtime option) while the box is there.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_call31
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
line() { python - <<PY
import json
try:
    d = json.loads(open("$O/$1.json").read().strip().splitlines()[-1])
    k = (d["roofline"].get("live_isolated") or d["roofline"]["live"])["kernel_ms_per_spp"]
    print("$1: %.1f Mrays/s %.4f ms/spp | alone: %s" % (d["value"], d["ms_per_spp"], k))
except Exception as e:
    print("$1: FAILED", e)
PY
}
B="--steps 3 --no-cpu-baseline --per-frame-frames 0"
python bench.py $B > $O/bench_base1.json 2>> $O/bench.err; el $(line bench_base1)
python bench.py $B --trace-tune $(( 40 | (8 << 8) )) > $O/bench_tune_40_8.json 2>> $O/bench.err; el $(line bench_tune_40_8)
python bench.py $B --trace-tune $(( 28 | (8 << 8) )) > $O/bench_tune_28_8.json 2>> $O/bench.err; el $(line bench_tune_28_8)
python bench.py $B --trace-tune $(( 32 | (12 << 8) )) > $O/bench_tune_32_12.json 2>> $O/bench.err; el $(line bench_tune_32_12)
cp raytracing_amd/librt_hip.so /tmp/librt_hip_base.so
cp raytracing_amd/variants/pkfma/librt_hip.so raytracing_amd/librt_hip.so
timeout 600 python -m pytest tests/test_gpu_headline_parity.py tests/test_gpu_parity.py -q -m gpu -x -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror|FAILED" | tail -3 > $O/pytest_variant.log; el variant tests: $(tail -1 $O/pytest_variant.log)
( RT_FUZZ_SEEDS=600 timeout 300 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -n 32 -p no:cacheprovider 2>&1 | grep -aE "passed|failed|rror|Timeout" | tail -2 ) > $O/fuzz_variant.log 2>&1; el variant fuzz: $(tail -1 $O/fuzz_variant.log)
python bench.py $B > $O/bench_pkfma1.json 2>> $O/bench.err; el $(line bench_pkfma1)
python bench.py $B > $O/bench_pkfma2.json 2>> $O/bench.err; el $(line bench_pkfma2)
cp /tmp/librt_hip_base.so raytracing_amd/librt_hip.so
python bench.py $B > $O/bench_base2.json 2>> $O/bench.err; el $(line bench_base2)
tail -2 $O/bench.err
el all done
