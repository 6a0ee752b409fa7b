#!/bin/bash
# Round 4, call 16: two node fetches in flight per lane priced on the bare visit chain: the same visit
# twice per pass with both fetches issued together, against one chain per lane, at 1 .. 26 waves per CU.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_call16
mkdir -p $O
cd $R
timeout 300 tools/bin/visit_mb 0.93 0.87 4096 > $O/visit_microbench_two_chains.json 2> $O/err.log
python - <<PY
import json
d = json.load(open("$O/visit_microbench_two_chains.json"))
by = {}
for r in d["runs"]:
    by.setdefault(r["kernel"], []).append((r["waves_per_cu"], round(r["gvisits_per_s"], 1)))
for k, v in by.items():
    print(k, v)
PY
