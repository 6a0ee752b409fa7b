"""profiles/r02_trace_counters.json from the rocprofv3 --pmc passes of tools/pmc_bench2.sh: per kernel of the
hot path, the busy fraction of every unit it could be bound by, each against a ceiling CALIBRATED with
tools/issue_microbench.hip on the same counters, plus HBM traffic.  bench.py reads the file for `roofline`.

  VALU   VALUBusy (= SQ_ACTIVE_INST_VALU x 4 / SIMDs / cycles) counts 4 cycles per wave64 instruction, but the
         gfx950 SIMD retires v_fma/mul/add/mov/and-class instructions in ~2.3 cycles: kernels of pure 4-cycle
         instructions saturate at 0.96, of pure 2-cycle ones at 1.72 (profiles/r02_issue_microbench_pmc.txt).
         For a kernel whose static opcode mix has a fraction f of 2-cycle instructions (tools/isa_mix.py) the
         ceiling is 4 / (f * 4 / 1.72 + (1 - f) * 4 / 0.96).
  SALU   SALUBusy saturates at 0.96 per CU x 4 (one scalar issue per SIMD per 4 cycles): reported against 1.0.
  L1/TA  TA_TA_BUSY_sum / TAs / cycles: 0.99 for saturating divergent dwordx4 loads.
  HBM    read bytes = 32 x TCC_EA0_RDREQ_32B + 64 x TCC_EA0_RDREQ_64B + 128 x TCC_EA0_RDREQ_128B (pass `rdreq`; round 6: checked against known byte
         counts per access pattern, profiles/r06_fetch_size_calibration.json -- this rocprofv3's FETCH_SIZE tallies every request at 64 bytes, i.e. HALF
         of a coalesced 16 B / lane stream or of a 128-byte record gather, and is right only for 64-byte records), + WRITE_SIZE KiB (exact for 16- and
         12-byte-per-lane streams), per launch / duration against 8 TB/s.  Without the `rdreq` pass: FETCH_SIZE x 0.99 (rounds 1 - 5; `hbm_read_how` says which).
  The fraction f of a TRACE kernel is its DYNAMIC mix: per-loop static opcode counts weighted by how often each loop's body runs
  (tools/isa_mix.py --loops, pass counts from tools/wave_schedule_model.py); k_shade has no loops worth weighting (static mix).
  The file records the SHA-256 of the code object the counters were collected from (raytracing_amd/codeobj.py):
  bench.py marks `roofline.stale` when the library it runs is another one.
usage: python tools/make_counters_json.py <pmc dir> <config> <out.json> closest=<f2> shadow=<f2> shade=<f2>
       (values as printed by tools/isa_mix.py; run on the tree the counters were collected from)"""
import csv, glob, json, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import codeobj

d, config, out = sys.argv[1], sys.argv[2], sys.argv[3]
mix = dict(a.split("=") for a in sys.argv[4:])
KERNELS = {"closest": "k_trace_w4<false", "shadow": "k_trace_w4<true", "shade": "k_shade<"}
CUS, SIMDS = 256, 1024


# Launches that count: those of at least 3 % of the kernel's longest launch in the same pass.  (RT_CTX_OPT_ADAPTIVE_FOLD's probe frame puts
# a few dozen launches of ~32 K paths in front of the frame's own of ~265 M: they would not move a busy FRACTION, but they would deflate every
# per-launch mean by their share of the launch COUNT.  The frame's smallest launch, the last bounce, is above a tenth of its largest.)
MIN_SHARE = 0.03


def mean(passname, counter, match):
    vals = []
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (d, passname), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and match in r["Kernel_Name"]]
        if not rows:
            continue
        dur = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in rows]
        vals += [float(r["Counter_Value"]) for r, t in zip(rows, dur) if t >= MIN_SHARE * max(dur)]
    return sum(vals) / len(vals) if vals else None


def launches(match):
    for f in glob.glob("%s/stats/**/*kernel_trace.csv" % d, recursive=True):
        dur = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if match in r["Kernel_Name"]]
        dur = [t for t in dur if t >= MIN_SHARE * max(dur)]
        if dur:
            return len(dur), sum(dur) / len(dur) * 1e-6
    for f in glob.glob("%s/stats/**/*kernel_stats.csv" % d, recursive=True):      # the summary alone (the trace was too large to keep)
        for r in csv.DictReader(open(f)):
            if match in r["Name"]:
                return int(r["Calls"]), float(r["AverageNs"]) * 1e-6
    return None, None


try:
    doc = json.load(open(out))
except Exception:
    doc = {}
entry = {}
for key, match in KERNELS.items():
    cyc = mean("sq", "GRBM_GUI_ACTIVE", match) / 8.0          # summed over the 8 XCDs
    calls, avg_ms = launches(match)
    f2 = float(mix.get(key, 0.5))
    valu_raw = mean("busy", "SQ_ACTIVE_INST_VALU", match) * 4.0 / SIMDS / (mean("busy", "GRBM_GUI_ACTIVE", match) / 8.0)
    valu_ceiling = 1.0 / (f2 / 1.72 + (1.0 - f2) / 0.96)
    salu_raw = mean("busy", "SQ_ACTIVE_INST_SCA", match) * 4.0 / SIMDS / (mean("busy", "GRBM_GUI_ACTIVE", match) / 8.0)
    ta = mean("ta", "TA_TA_BUSY_sum", match) / CUS / (mean("ta", "GRBM_GUI_ACTIVE", match) / 8.0)
    fetch, write = mean("fetch", "FETCH_SIZE", match), mean("write", "WRITE_SIZE", match)
    sized = [mean("rdreq", "TCC_EA0_RDREQ_%s_sum" % n, match) for n in ("32B", "64B", "128B")]
    if all(v is not None for v in sized):
        read_bytes, read_how = 32.0 * sized[0] + 64.0 * sized[1] + 128.0 * sized[2], "32 / 64 / 128-byte read requests counted separately (pass rdreq)"
    else:
        read_bytes, read_how = fetch * 0.99 * 1024.0, "FETCH_SIZE x 0.99 (every request tallied at 64 bytes: a lower bound where 128-byte requests occur)"
    hbm_bytes = read_bytes + write * 1024.0
    acc = mean("tcp", "TCP_TOTAL_CACHE_ACCESSES_sum", match)
    l2req = mean("tcp", "TCP_TCC_READ_REQ_sum", match)
    entry[key] = {
        "kernel": match + "...>", "launches_profiled": calls, "avg_launch_ms": avg_ms, "cycles_per_launch": cyc,
        "valu_busy_raw": valu_raw, "valu_fast_opcode_fraction": f2, "valu_ceiling_raw": valu_ceiling, "valu_busy": valu_raw / valu_ceiling,
        "salu_busy": salu_raw / 0.96, "l1_ta_busy": ta / 0.99,
        "hbm_read_bytes_per_launch": read_bytes, "hbm_write_bytes_per_launch": write * 1024.0, "hbm_read_how": read_how,
        "fetch_size_KiB_per_launch": fetch, "read_requests_per_launch": dict(zip(("32B", "64B", "128B"), sized)) if all(v is not None for v in sized) else None,
        "hbm_bytes_per_launch": hbm_bytes, "hbm_GBs": hbm_bytes / (avg_ms * 1e-3) / 1e9, "hbm_frac": hbm_bytes / (avg_ms * 1e-3) / 8e12,
        "per_launch": {"valu_instructions": mean("sq", "SQ_INSTS_VALU", match), "salu_instructions": mean("sq", "SQ_INSTS_SALU", match),
                       "vmem_instructions": mean("sq", "SQ_INSTS_VMEM", match), "lds_instructions": mean("sq", "SQ_INSTS_LDS", match),
                       "l1_accesses": acc, "l2_requests": l2req, "l1_hit_rate": 1.0 - l2req / acc,
                       "l2_hit_rate": mean("tcc", "TCC_HIT_sum", match) / mean("tcc", "TCC_REQ_sum", match)},
    }
doc["config_%s" % config] = entry
doc["_how"] = "tools/pmc_bench2.sh + tools/make_counters_json.py (see its docstring for every ceiling); VALU mix: tools/isa_mix.py --loops"
doc["_code_object_sha256"] = codeobj.code_object_sha256()
doc["_fold"] = os.environ.get("RT_COUNTERS_FOLD", "surface area")      # which fold of the trees the profiled runs walked (bench.py --adaptive-fold)
json.dump(doc, open(out, "w"), indent=1)
for k, e in entry.items():
    print(k, {n: round(e[n], 4) for n in ("valu_busy", "salu_busy", "l1_ta_busy", "hbm_frac", "avg_launch_ms")})
