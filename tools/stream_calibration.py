"""profiles/r06_fetch_size_calibration.json from the counter passes over tools/stream_microbench.hip (known byte counts per access pattern):
what rocprofv3's FETCH_SIZE / WRITE_SIZE report on gfx950 against the truth, and the sized-request counters that ARE the truth.
usage: python tools/stream_calibration.py <dir with cal_fetch / cal_write / cal_sized passes and stream_mb.log> <out.json>"""
import csv, glob, json, re, sys, collections

d, out = sys.argv[1], sys.argv[2]
known = {}
for line in open(d + "/stream_mb.log"):
    m = re.match(r"(\S+)\s+bytes (\d+)\s+ms ([\d.]+)\s+TB/s ([\d.]+)", line)
    if m:
        known[m.group(1)] = dict(bytes=float(m.group(2)), ms=float(m.group(3)), TBs=float(m.group(4)))
names = {"k_read16": "k_read16", "k_write16": "k_write16", "k_write12": "k_write12", "k_read12": "k_read12", "k_rand<4>": "k_rand64", "k_rand<8>": "k_gather128"}


def counters(passname):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (d, passname), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if k in names:
                acc[names[k]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


fetch, write, sized = counters("cal_fetch"), counters("cal_write"), counters("cal_sized")
doc = {"_how": "tools/stream_microbench.hip (every kernel moves a known byte count, 2 GiB buffers, past the Infinity Cache) under rocprofv3 --pmc, one pass per "
               "counter group; tools/stream_calibration.py", "patterns": {}}
for k, v in known.items():
    e = dict(known_bytes=v["bytes"], TBs=v["TBs"])
    reads = k in ("k_read16", "k_read12", "k_rand64", "k_gather128")
    if reads and k in fetch:
        e["FETCH_SIZE_bytes"] = fetch[k]["FETCH_SIZE"] * 1024.0
        e["FETCH_SIZE_over_known"] = round(e["FETCH_SIZE_bytes"] / v["bytes"], 4)
        e["read_requests"] = fetch[k].get("TCC_EA0_RDREQ_sum")
    if reads and k in sized:
        s = sized[k]
        b = 32.0 * s.get("TCC_EA0_RDREQ_32B_sum", 0.0) + 64.0 * s.get("TCC_EA0_RDREQ_64B_sum", 0.0) + 128.0 * s.get("TCC_EA0_RDREQ_128B_sum", 0.0)
        e["sized_requests"] = {n: s.get("TCC_EA0_RDREQ_%s_sum" % n) for n in ("32B", "64B", "128B")}
        e["sized_request_bytes_over_known"] = round(b / v["bytes"], 4)
    if not reads and k in write:
        e["WRITE_SIZE_bytes"] = write[k]["WRITE_SIZE"] * 1024.0
        e["WRITE_SIZE_over_known"] = round(e["WRITE_SIZE_bytes"] / v["bytes"], 4)
    doc["patterns"][k] = e
json.dump(doc, open(out, "w"), indent=1)
for k, e in doc["patterns"].items():
    print(k, {n: e[n] for n in e if n.endswith("over_known") or n == "TBs"})
