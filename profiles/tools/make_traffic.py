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
ap.add_argument("--chunk", type=int, default=250_000_000)  # bench.py's default
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
# FETCH_SIZE is 64 B x fabric read requests on gfx950, and a request for both sectors of a 128-byte L2 line is ONE
# request: streams of >= 8 B per lane and half of the text-record runs are counted at half their bytes.  The
# correction is per access class (profiles/tools/traffic_classes.py: three builds of the kernel under the counter,
# factors from known byte counts in profiles/r04_fetch_calibration.json) and comes out as one factor for this
# kernel on this workload's kind of reads.
cal, factor = None, 1.0
try:
    cal_path = os.path.join("profiles", "r09_traffic_classes.json")          # (this round's kernel; round 4's otherwise)
    if not os.path.exists(cal_path):
        cal_path = os.path.join("profiles", "r04_traffic_classes.json")
    cal = json.load(open(cal_path))
    factor = float(cal["read_factor_corrected_over_FETCH_SIZE"])
except (OSError, KeyError, ValueError):
    cal = None
wk = {"draft_mbp": a.draft_mbp, "pairs_per_launch": min(a.chunk, a.pairs), "k": a.k}
if a.repeats:
    wk["repeats"] = True
out = {"workload": wk,
       "kernel_build_id": bench.kernel_build_id(),
       "FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w,
       "hbm_bytes_per_launch_uncalibrated": (f + w) * 1024.0,
       "hbm_bytes_per_launch": (f * factor + w) * 1024.0,
       "traffic_calibration": {
           "read_factor": factor,
           "source": (cal_path + " (per-pair bytes by class: " +
                      json.dumps({k: round(v, 1) for k, v in cal["per_pair_bytes_corrected"].items()}) +
                      " corrected, " + json.dumps({k: round(v, 1) for k, v in cal["per_pair_bytes_counted"].items()}) +
                      " as FETCH_SIZE counts them); class factors from profiles/r04_fetch_calibration.json, "
                      "profiles/r04j_gather_width2.txt") if cal else "none found: FETCH_SIZE x 1",
           "reads": "FETCH_SIZE x read_factor (the read stream at its known size, 32-byte probes x 1, runs of text records x 4/3)",
           "writes": "WRITE_SIZE x 1 (coalesced 4-byte stores: exact in the calibration)"},
       "TCC_MISS_per_launch": miss,
       "dispatches_profiled": dispatches_of(f"gpurun_out/{tag}_pmc_FETCH_SIZE.csv", "FETCH_SIZE"),
       "source": f"profiles/{tag}_pmc_FETCH_SIZE.csv + {tag}_pmc_WRITE_SIZE.csv (rocprofv3 --pmc, separate passes)"}
json.dump(out, open(f"gpurun_out/traffic_{tag.split('_')[0]}.json", "w"), indent=1)
print(json.dumps(out))
