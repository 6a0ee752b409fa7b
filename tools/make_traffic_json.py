"""profiles/trace_closest_hbm_traffic.json from the rocprofv3 --pmc passes of tools/pmc_bench.sh:
mean FETCH_SIZE / WRITE_SIZE per k_trace<closest> dispatch (all dispatches of the run have the
same size when warm-up and timed batches hold the same number of samples), plus the FETCH_SIZE
calibration on this kernel's access pattern (random 64-byte records, known byte count)."""
import csv, collections, json, sys

def per_kernel(path, counter, match):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if r["Counter_Name"] == counter and match in r["Kernel_Name"]]
    return vals

d, config, out = sys.argv[1], sys.argv[2], sys.argv[3]
fetch = per_kernel(d + "/fetch_counter_collection.csv", "FETCH_SIZE", "k_trace<false")
write = per_kernel(d + "/write_counter_collection.csv", "WRITE_SIZE", "k_trace<false")
calib = per_kernel(d + "/calib_counter_collection.csv", "FETCH_SIZE", "k_fetch<0>")
known = 256 * 16 * 64 * 200 * 64.0            # blocks x lanes x iterations x 64-byte records (tools/fetch_microbench.hip)
factor = known / (calib[0] * 1024.0) if calib else None
try:
    doc = json.load(open(out))
except Exception:
    doc = {}
valu = per_kernel(d + "/sq_counter_collection.csv", "SQ_INSTS_VALU", "k_trace<false")
gui = per_kernel(d + "/sq_counter_collection.csv", "GRBM_GUI_ACTIVE", "k_trace<false")
tcp = per_kernel(d + "/tcp_counter_collection.csv", "TCP_TOTAL_CACHE_ACCESSES_sum", "k_trace<false")
hit = per_kernel(d + "/tcc_counter_collection.csv", "TCC_HIT_sum", "k_trace<false")
req = per_kernel(d + "/tcc_counter_collection.csv", "TCC_REQ_sum", "k_trace<false")
l2r = per_kernel(d + "/tcp_counter_collection.csv", "TCP_TCC_READ_REQ_sum", "k_trace<false")
cycles = sum(gui) / 8.0                        # GRBM_GUI_ACTIVE is summed over the 8 XCDs
doc["config_%s" % config] = {
    # 256 CUs x 4 SIMDs, one wave64 VALU instruction occupies a SIMD for 4 cycles
    "valu_issue_utilisation": sum(valu) * 4.0 / 1024.0 / cycles,
    "l1_accesses_per_clk_per_cu": sum(tcp) / 256.0 / (sum(gui) / 8.0) * (len(gui) / len(tcp)),
    "l1_hit_rate": 1.0 - sum(l2r) / sum(tcp),
    "l2_hit_rate": sum(hit) / sum(req),
    "kernel": next(r["Kernel_Name"].split("(")[0].replace("void ", "") for r in csv.DictReader(open(d + "/fetch_counter_collection.csv"))
                   if "k_trace<false" in r["Kernel_Name"]),
    "dispatches": len(fetch),
    "FETCH_SIZE_KB_per_dispatch": sum(fetch) / len(fetch),
    "WRITE_SIZE_KB_per_dispatch": sum(write) / len(write),
    "calibration": {"pattern": "random 64-byte records out of a 3.8 GB table, 4 x global_load_dwordx4 per lane",
                    "known_bytes": known, "FETCH_SIZE_KB": calib[0] if calib else None,
                    "bytes_per_reported_byte": factor},
    "bytes_per_launch": (sum(fetch) / len(fetch) * (factor or 1.0) + sum(write) / len(write)) * 1024.0,
    "note": "FETCH_SIZE x the calibrated factor for this access pattern (the guide's x2 applies to wide coalesced "
            "streams; random 64-byte record fetches calibrate to ~1.0) + WRITE_SIZE (uncalibrated, 16 B per ray).",
}
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps(doc["config_%s" % config], indent=1))
