"""profiles/r04_traffic_classes.json: the hot kernel's HBM reads split into access classes, each corrected by what
profiles/tools/fetch_calib.hip / gather_width2.hip measured on known byte counts (profiles/r04_fetch_calibration.json,
r04j_gather_width2.txt):
  * FETCH_SIZE = 64 B x fabric read requests (TCC_EA0_RDREQ); a request is 64 B (one sector of a 128-byte L2 line) or,
    when both sectors of a line are asked for together, 128 B -- which FETCH_SIZE still counts as 64;
  * streams of >= 8 B per lane: every request 128 B                  -> true bytes = 2 x FETCH (= the known byte count)
  * streams of 4 B per lane (and narrower): 64-byte requests         -> x 1
  * a random aligned 32-byte probe: 1.04 requests of 64 B            -> x 1 (64 B fetched for 32 B used)
  * a run of five 16-byte records at a random place: 2 sectors, half of the time in one line (one 128 B request)
                                                                      -> true bytes = 4/3 x FETCH
Three builds of the kernel under rocprofv3 --pmc FETCH_SIZE on the same 20 M pairs: the tree's, one without text
records (-DARKS_CAL_NO_TREC) and one without probes (-DARKS_CAL_NO_PROBE: no diagonals either, so the read stream
alone, whose bytes are known exactly).  FETCH(full) - FETCH(no records) = the records; FETCH(no records) -
FETCH(no probes) = the probes; the stream is taken at its known size.
usage: traffic_classes.py <dir> <out.json>   (<dir>/{full,notrec,noprobe}/ = rocprofv3 outputs, <dir>/full.log = ab.py's stdout)"""
import csv, glob, json, os, sys

src, out_path = sys.argv[1], sys.argv[2]


def fetch(sub):
    vals = []
    for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "map_reads_s_kernel" in r.get("Kernel_Name", "") and r["Counter_Name"] == "FETCH_SIZE":
                vals.append(float(r["Counter_Value"]) * 1024.0)
    return sum(vals) / len(vals), len(vals)


st = {}
for ln in open(os.path.join(src, "full.log")):
    if ln.startswith("stream_bytes"):
        st = {k: int(v) for k, v in (kv.split("=") for kv in ln.split()[1:])}
full, n1 = fetch("full")
notrec, n2 = fetch("notrec")
noprobe, n3 = fetch("noprobe")
pairs = st["reads"] // 2
stream_known = st["codes"] + st["word_off"] + st["lens"] + st["eval"]          # (N masks: < 1 % of the tiles fetch them)
records_counted = full - notrec
probes_counted = notrec - noprobe
corrected = stream_known + probes_counted + records_counted * 4.0 / 3.0
res = {
    "pairs_per_launch": pairs, "dispatches": [n1, n2, n3],
    "FETCH_SIZE_bytes": {"full": full, "no_text_records": notrec, "no_probes_no_records": noprobe},
    "per_pair_bytes_counted": {"stream": noprobe / pairs, "probes": probes_counted / pairs, "text_records": records_counted / pairs,
                               "all": full / pairs},
    "per_pair_bytes_corrected": {"stream": stream_known / pairs, "probes": probes_counted / pairs,
                                 "text_records": records_counted * 4.0 / 3.0 / pairs, "all": corrected / pairs},
    "stream_known_bytes": stream_known, "stream_factor_known_over_counted": stream_known / noprobe,
    "read_factor_corrected_over_FETCH_SIZE": corrected / full,
    "method": __doc__.split("usage:")[0].strip(),
}
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "method"}, indent=1))
