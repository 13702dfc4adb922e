"""profiles/r04_fetch_calibration.json: what rocprofv3's FETCH_SIZE / WRITE_SIZE report on gfx950 against KNOWN byte counts,
per access class of map_reads_s_kernel ("next" 4).
usage: fetch_calib.py <dir with the rocprofv3 outputs of scratch/run_calib.sh> <out.json>
  <dir>/micro_<COUNTER>/   rocprofv3 --pmc <COUNTER> -- fetch_calib        (+ micro.log = its stdout: the known bytes)
  <dir>/noprobe_<COUNTER>/ rocprofv3 --pmc <COUNTER> -- ab.py cal_noprobe  (+ noprobe.log: stream_bytes ...)"""
import csv, glob, json, os, re, sys

src, out_path = sys.argv[1], sys.argv[2]


def counters(sub):
    """{kernel name prefix: {counter: [values per dispatch]}}"""
    res = {}
    for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "").split("(")[0]
            res.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"] or 0))
    return res


known = {}
for ln in open(os.path.join(src, "micro.log")):
    m = re.match(r"kernel=(.+?) requested_bytes=(\d+) lines64_bytes=(\d+)", ln)
    if m:
        known[m.group(1)] = (int(m.group(2)), int(m.group(3)))
micro = {}
for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_EA0_RDREQ_sum", "TCC_HIT_sum", "TCC_EA0_WRREQ_sum"):
    for k, v in counters("micro_" + c).items():
        for name, vals in v.items():
            micro.setdefault(k, {})[name] = sum(vals) / len(vals)
classes = {}
for kern, (req, lines) in known.items():
    hit = [k for k in micro if kern.split("<")[0] in k and (("<" not in kern) or kern.split("<")[1].rstrip(">") in k)]
    if not hit:
        continue
    c = micro[hit[0]]
    e = {"requested_bytes": req, "bytes_of_the_64B_lines_touched": lines, "counters_per_launch": c}
    if kern.startswith("write4"):
        if c.get("WRITE_SIZE"):
            e["WRITE_SIZE_bytes"] = c["WRITE_SIZE"] * 1024
            e["factor_requested_over_WRITE_SIZE"] = req / (c["WRITE_SIZE"] * 1024)
    elif c.get("FETCH_SIZE"):
        if "TCC_EA0_RDREQ_128B_sum" in c:
            # requests by size: what the fabric was asked for, whatever FETCH_SIZE makes of it
            by = 32 * c.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * c.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * c["TCC_EA0_RDREQ_128B_sum"]
            e["bytes_by_request_size"] = by
            e["factor_lines_over_bytes_by_request_size"] = lines / by if by else None
        e["FETCH_SIZE_bytes"] = c["FETCH_SIZE"] * 1024
        e["factor_requested_over_FETCH_SIZE"] = req / (c["FETCH_SIZE"] * 1024)
        e["factor_lines_over_FETCH_SIZE"] = lines / (c["FETCH_SIZE"] * 1024)
    classes[kern] = e
res = {"microbenchmarks": classes, "source": "profiles/tools/fetch_calib.hip under rocprofv3 --pmc, one counter group per pass; 8 GiB table"}
# the hot kernel's own read stream: the build without probes
try:
    st = {}
    for ln in open(os.path.join(src, "noprobe.log")):
        if ln.startswith("stream_bytes"):
            st = {k: int(v) for k, v in (kv.split("=") for kv in ln.split()[1:])}
    nop = counters("noprobe_FETCH_SIZE")
    hot = [k for k in nop if "map_reads_s_kernel" in k]
    vals = nop[hot[0]]["FETCH_SIZE"]
    fetch = sum(vals) / len(vals) * 1024
    # 8-byte words, offsets, lengths, eval bytes of every read; N masks of the reads that may hold N (a lower and an
    # upper bound: masks are fetched per tile -- a tile with one such read fetches all of its masks)
    rq = counters("noprobe_RDREQ")
    hot2 = [k for k in rq if "map_reads_s_kernel" in k]
    by = None
    if hot2:
        m = {n: sum(v) / len(v) for n, v in rq[hot2[0]].items()}
        by = 32 * m.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * m.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * m.get("TCC_EA0_RDREQ_128B_sum", 0)
    base = st["codes"] + st["word_off"] + st["lens"] + st["eval"]
    lo = base + 4 * (st["codes"] // 8) * st["reads_whose_masks_may_be_fetched"] // max(st["reads"], 1)
    res["hot_kernel_without_probes"] = {
        "known_stream_bytes_without_masks": base, "known_stream_bytes_with_masks_of_flagged_reads": lo, "stream": st,
        "FETCH_SIZE_bytes": fetch, "dispatches": len(vals), "factor_known_over_FETCH_SIZE": base / fetch,
        "bytes_by_request_size": by, "factor_known_over_bytes_by_request_size": (base / by) if by else None,
        "source": "profiles/tools/ab.py on a -DARKS_CAL_NO_PROBE build (no probe -> no diagonal -> no text record), AB_EVAL=1"}
except (OSError, KeyError, IndexError) as e:
    res["hot_kernel_without_probes"] = {"error": repr(e)}
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps(res, indent=1))
