"""profiles/traffic_<round>.json from the FETCH_SIZE / WRITE_SIZE passes of profiles/prof.sh: HBM bytes per
launch of the dominant kernel, stamped with the workload and the id of the kernel build the counters were
taken on (bench.py reports `roofline.traffic` only when both match what it is running).
usage: make_traffic.py <tag> [bench args of the profiled run...]"""
import argparse, csv, json, os, sys
sys.path.insert(0, os.getcwd())
import bench

tag = sys.argv[1]
ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=500_000_000)
ap.add_argument("--chunk", type=int, default=100_000_000)  # bench.py's default
ap.add_argument("--draft-mbp", type=float, default=3000.0)
ap.add_argument("--k", type=int, default=60)
ap.add_argument("--repeats", action="store_true")
a, _ = ap.parse_known_args(sys.argv[2:])


def mean_of(path, counter):
    best = None
    for r in csv.DictReader(open(path)):
        # the hot map kernel without the -v counters (second template argument false): seed tile kernel, or the
        # minimizer tile kernel (FULL = false)
        k = r["kernel"]
        hot = ("map_reads_s_kernel<" in k and ", false, " in k.split("map_reads_s_kernel<")[1][:12]) or \
              ("map_reads_b_kernel<" in k and "false, false" in k)
        if r["counter"] == counter and hot:
            v = float(r["mean_per_dispatch"])
            best = v if best is None else max(best, v)
    return best


def dispatches_of(path, counter):
    """dispatches of the hot kernel the means are taken over"""
    best = 0
    for r in csv.DictReader(open(path)):
        k = r["kernel"]
        hot = ("map_reads_s_kernel<" in k and ", false, " in k.split("map_reads_s_kernel<")[1][:12]) or \
              ("map_reads_b_kernel<" in k and "false, false" in k)
        if r["counter"] == counter and hot:
            best = max(best, int(r["dispatches"]))
    return best


f = mean_of(f"gpurun_out/{tag}_pmc_FETCH_SIZE.csv", "FETCH_SIZE")
w = mean_of(f"gpurun_out/{tag}_pmc_WRITE_SIZE.csv", "WRITE_SIZE")
try:
    miss = mean_of(f"gpurun_out/{tag}_pmc_TCC_HIT_sum_TCC_MISS_sum.csv", "TCC_MISS_sum")
except OSError:
    miss = None
# read requests by size (one pass of four TCC counters): what the L2s asked of the fabric, byte for byte.  FETCH_SIZE
# tallies every request at 64 bytes on gfx950 -- 128-byte requests (two adjacent lines: 8- and 16-byte-per-lane
# streams, runs of text records) count half, as profiles/tools/fetch_calib.hip shows on known byte counts
# (profiles/r04_fetch_calibration.json); 32-byte probes cost a 64-byte request each and are counted as such.
rq = f"gpurun_out/{tag}_pmc_TCC_EA0_RDREQ_sum_TCC_EA.csv"
by_size = None
try:
    r32, r64, r128, rall = (mean_of(rq, "TCC_EA0_RDREQ_32B_sum"), mean_of(rq, "TCC_EA0_RDREQ_64B_sum"),
                            mean_of(rq, "TCC_EA0_RDREQ_128B_sum"), mean_of(rq, "TCC_EA0_RDREQ_sum"))
    if r128 is not None and rall:
        by_size = {"requests": rall, "requests_32B": r32 or 0.0, "requests_64B": r64 or 0.0, "requests_128B": r128,
                   "read_bytes": 32.0 * (r32 or 0.0) + 64.0 * (r64 or 0.0) + 128.0 * r128}
except OSError:
    pass
wk = {"draft_mbp": a.draft_mbp, "pairs_per_launch": min(a.chunk, a.pairs), "k": a.k}
if a.repeats:
    wk["repeats"] = True
out = {"workload": wk,
       "kernel_build_id": bench.kernel_build_id(),
       "FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w,
       "hbm_bytes_per_launch_uncalibrated": (f + w) * 1024.0,
       "hbm_bytes_per_launch": ((by_size["read_bytes"] if by_size else f * 1024.0) + w * 1024.0),
       "read_requests_by_size": by_size,
       "traffic_calibration": {
           "source": "profiles/r04_fetch_calibration.json (profiles/tools/fetch_calib.hip + a build of the hot kernel without "
                     "probes, known byte counts against rocprofv3)",
           "reads": ("32 B x TCC_EA0_RDREQ_32B + 64 B x TCC_EA0_RDREQ_64B + 128 B x TCC_EA0_RDREQ_128B per launch"
                     if by_size else "FETCH_SIZE x 1 (the by-size counters were not collected)"),
           "FETCH_SIZE_over_read_bytes": (f * 1024.0 / by_size["read_bytes"]) if by_size else None,
           "writes": "WRITE_SIZE x 1 (coalesced 4-byte stores: exact in the calibration)"},
       "TCC_MISS_per_launch": miss,
       "dispatches_profiled": dispatches_of(f"gpurun_out/{tag}_pmc_FETCH_SIZE.csv", "FETCH_SIZE"),
       "source": f"profiles/{tag}_pmc_FETCH_SIZE.csv + {tag}_pmc_WRITE_SIZE.csv (rocprofv3 --pmc, separate passes)"}
json.dump(out, open(f"gpurun_out/traffic_{tag.split('_')[0]}.json", "w"), indent=1)
print(json.dumps(out))
