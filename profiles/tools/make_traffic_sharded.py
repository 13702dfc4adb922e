"""profiles/traffic_sharded_<tag>.json from the FETCH_SIZE / WRITE_SIZE passes of profiles/prof_sharded.sh: the HBM
bytes of ONE step of `bench.py --sharded-index` (all the read pairs through bucket, probe, map_reads_s_kernel<REMOTE>
and the pair rule, on every local rank), kernel by kernel and summed, stamped with the workload and the kernel build.
FETCH_SIZE is taken x 1 here (no per-class calibration for these kernels): streams of >= 8 B per lane are counted at
half their bytes on gfx950, so the sum is a LOWER bound of the traffic and of roofline.frac.
usage: make_traffic_sharded.py <tag> <steps> <warmup> [bench args of the counter passes...]"""
import argparse, csv, json, os, sys
sys.path.insert(0, os.getcwd())
import bench

tag, steps, warmup = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=500_000_000)
ap.add_argument("--draft-mbp", type=float, default=3000.0)
ap.add_argument("--k", type=int, default=60)
ap.add_argument("--shards", type=int, default=8)
a, _ = ap.parse_known_args(sys.argv[4:])

# the kernels of a step; the map kernels of the timed passes are the instantiations WITHOUT counters (second template
# argument false), the counters pass in front of them runs the others
STEP = ("seed_bucket_kernel", "seeds_probe_segs_kernel", "map_reads_s_kernel<", "map_reads_b_kernel<", "map_reads_kernel<",
        "pair_gate_kernel", "pairs_kernel")


def per_step(path, counter):
    out = {}
    for r in csv.DictReader(open(path)):
        k = r["kernel"]
        name = next((s for s in STEP if "arks::" + s in k), None)
        if name is None or r["counter"] != counter:
            continue
        is_map = name.endswith("<")
        if is_map:
            targs = k.split(name)[1].split(">")[0].split(",")
            if len(targs) < 2 or targs[1].strip() != "false":
                continue                      # the counters pass's instantiation
        passes = (warmup + steps) if is_map else (1 + warmup + steps)
        key = k.split("(")[0].replace("void ", "")
        out[key] = out.get(key, 0.0) + float(r["sum"]) / passes     # KB per step
    return out


f = per_step(f"gpurun_out/{tag}_sharded_pmc_FETCH_SIZE.csv", "FETCH_SIZE")
w = per_step(f"gpurun_out/{tag}_sharded_pmc_WRITE_SIZE.csv", "WRITE_SIZE")
kernels = sorted(set(f) | set(w))
out = {"workload": {"draft_mbp": a.draft_mbp, "pairs": a.pairs, "k": a.k, "shards": a.shards, "n_gpus": 1},
       "kernel_build_id": bench.kernel_build_id(),
       "per_kernel_GB_per_step": {k: {"fetch": f.get(k, 0.0) * 1024 / 1e9, "write": w.get(k, 0.0) * 1024 / 1e9} for k in kernels},
       "hbm_bytes_per_step_uncalibrated": (sum(f.values()) + sum(w.values())) * 1024.0,
       "note": "FETCH_SIZE x 1 + WRITE_SIZE, every kernel of the step on every local rank, per pass over all the pairs; "
               "a lower bound (64 B per fabric request: whole-line requests of the streams count half)",
       "source": f"profiles/{tag}_sharded_pmc_FETCH_SIZE.csv + {tag}_sharded_pmc_WRITE_SIZE.csv (rocprofv3 --pmc, separate passes, "
                 f"bench.py --sharded-index --steps {steps} --warmup {warmup})"}
json.dump(out, open(f"gpurun_out/traffic_sharded_{tag}.json", "w"), indent=1)
print(json.dumps(out, indent=1))
