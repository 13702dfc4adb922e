"""The at-scale end-to-end leg of bench.py alone, with the front end's stage times and ingest profile kept:
usage: e2e_scale_profile.py <out dir> [threads]   (ARKS_INGEST_PROFILE=1 ARKS_TIMING=1 are set for the CLI)"""
import json, os, sys
sys.path.insert(0, os.getcwd())
import torch
import bench
out = sys.argv[1]; os.makedirs(out, exist_ok=True)
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda", 0)
log = lambda m: print("[e2e]", m, file=sys.stderr, flush=True)
wl = bench.Workload(3000.0, 12_500_000, 12_500_000, 60, 0.55, dev, 0, log, want_stats=False, keep_draft=True)
os.environ["ARKS_INGEST_PROFILE"] = "1"
for tag, env in (("default", {}), ("pgzip", {"ARKS_PGZIP": "1"})):
    os.environ.update(env)
    os.environ["ARKS_BENCH_E2E_STDERR"] = os.path.join(out, f"cli_stderr_{tag}.txt")
    r = bench.end_to_end_scale(wl, dev, 0, log, {"pairs_per_s": 239403.0}, threads=threads)
    json.dump(r, open(os.path.join(out, f"e2e_{tag}.json"), "w"), indent=1)
    print(tag, r["gpu_cli"]["read_stage_ms"], r["gpu_cli"]["seconds"], r["gpu_cli"]["stage_ms"], flush=True)
