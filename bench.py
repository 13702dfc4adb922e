#!/usr/bin/env python3
"""bench.py -- throughput of the ARKS read->contig k-mer mapping hot path on MI355X.

One "step" = one pass of the hot path (pair gate -> per-read k-mer window keys -> contig k-mer
table probe -> per-read vote -> pair rule + (barcode, contig end) accumulation) over one resident
batch of synthetic linked-read pairs.  Workload at N=1: BASELINE.json configs[1] -- synthetic 50 Mbp
draft + 20 M linked-read pairs (R1 128 bp / R2 151 bp), k=60, j=0.55.  With N > 1 every rank holds a
replica of the index and its own 20 M pairs (the path shards over reads with no data-path
collective; weak scaling).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--pairs P] [--draft-mbp M]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import arcs_amd  # noqa: E402
from arcs_amd import synth  # noqa: E402

METRIC = "read k-mers hashed+probed/sec at k=60; 1/2/4/8 GPU; bit-exact .gv"
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def alg_bytes_per_window(k, bases, windows):
    """packed read stream + one key compare + one int value per window"""
    return 2.0 * bases / (8.0 * windows) + (2 * k + 7) // 8 + 4


def pmc_traffic(args):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary of this
    very workload (profiles/traffic_r*.json, written from separate --pmc passes by profiles/prof.sh);
    None when no summary matches the workload being run."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_r*.json"))):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        wl = d.get("workload", {})
        if wl.get("pairs_per_gpu") == args.pairs and wl.get("draft_mbp") == args.draft_mbp and wl.get("k") == args.k:
            best = d
    return best


def cpu_baseline(cs, batch, k, j, n_pairs_total, log):
    """the CPU oracle (a literal port of the reference path: per-window O(k) re-encode, exact
    hash map, ordered histogram; OpenMP over pairs) timed on the host cores, on a bounded sample
    of the same reads.  Test infrastructure, never the product path."""
    from oracle import pyoracle as O
    O.build_oracle()
    cores = os.cpu_count() or 1
    t0 = time.time()
    ox = O.OracleIndex(k).build(O.contig_ends(cs))
    log(f"cpu oracle index: {len(ox)} keys in {time.time() - t0:.1f}s")
    L = int(batch["lens"][0].item()) + int(batch["lens"][1].item())

    def run(n_pairs, threads):
        a = batch["ascii"][: n_pairs * L].cpu().numpy()
        lens = batch["lens"][: 2 * n_pairs].cpu().numpy().astype(np.uint32)
        offs = batch["offsets"][: 2 * n_pairs].cpu().numpy().astype(np.uint64)
        ok = batch["pair_ok"][:n_pairs].cpu().numpy()
        a = np.concatenate([a, np.zeros(1, np.uint8)])
        t = time.time()
        c, p, st = ox.map_pairs(a, offs, lens, j, pair_ok=ok, threads=threads)
        return time.time() - t, c, p, st

    probe = min(20000, n_pairs_total)
    dt, _, _, st = run(probe, cores)
    rate = st["windows"] / max(dt, 1e-9)
    n = int(min(n_pairs_total, max(probe, 15.0 * rate / (st["windows"] / probe))))
    dt, c, p, st = run(n, cores)
    return {"value": st["windows"] / dt, "unit": "k-mers/s", "cores": cores, "kind": "port",
            "sample": f"first {n} pairs ({st['windows']} windows) of the rank-0 batch, "
                      f"{dt:.1f}s, OpenMP {cores} threads, index resident in host RAM"}, (n, c, p, st)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=20_000_000, help="read pairs per GPU")
    ap.add_argument("--draft-mbp", type=float, default=50.0)
    ap.add_argument("--k", type=int, default=60)
    ap.add_argument("--j", type=float, default=0.55)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sharded-index", action="store_true",
                    help="BASELINE configs[3]: rank r holds shard r of the index, every rank maps the SAME "
                         "batch, one all-reduce(MAX) of the per-read votes per step (default: index replicas, "
                         "reads sharded, no data-path collective)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # one process per GPU; ARKS_BENCH_BACKEND=gloo lets several ranks share one GPU (smoke tests of
    # the multi-rank control flow on a single-GPU box; the driver's runs use nccl = RCCL)
    backend = os.environ.get("ARKS_BENCH_BACKEND", "nccl")
    local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local)
    red_dev = dev if backend == "nccl" else torch.device("cpu")

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    assert arcs_amd.device_count() >= 1, "libarks_hip sees no gfx950 device (no CPU fallback)"
    k, j = args.k, args.j

    # ---- draft + index replica -----------------------------------------------------------------
    t0 = time.time()
    contigs = synth.make_draft(int(args.draft_mbp * 1e6), seed=synth.SEED)
    cs = synth.contigs_to_strings(contigs)
    ends = arcs_amd.contig_ends(cs)
    log(f"draft: {len(contigs)} contigs, {sum(map(len, cs))} bp, {len(ends)} ends in {time.time() - t0:.1f}s")
    t0 = time.time()
    if args.sharded_index:
        index = arcs_amd.ArksIndex.build_shard(ends, k, rank, world, device=local)
    else:
        index = arcs_amd.ArksIndex.build(ends, k, device=local, want_stats=(rank == 0))
    log(f"index: {len(index)} keys, {index.device_bytes / 2**30:.2f} GiB, built in {time.time() - t0:.1f}s "
        f"{index.build_stats}")

    # ---- reads, resident in HBM before the timed region ------------------------------------------
    t0 = time.time()
    batch = synth.make_read_pairs(contigs, args.pairs, seed=synth.SEED + 1 + (0 if args.sharded_index else rank),
                                  device=dev)
    reads = arcs_amd.PackedReads.from_arrays_device(batch["ascii"], batch["offsets"], batch["lens"], device=local)
    windows = reads.windows(k)
    bases = int(batch["lens"].to(torch.int64).sum().item())
    log(f"reads: {args.pairs} pairs, {windows} windows, packed in {time.time() - t0:.1f}s")
    b_alg = alg_bytes_per_window(k, bases, windows)

    stats = torch.zeros(8, dtype=torch.int64, device=dev)
    stored = torch.zeros(1, dtype=torch.int64, device=dev)
    n_barcodes = int(batch["barcode_id"].max().item()) + 1
    imap = arcs_amd.ImapAccumulator(max(1 << 16, 8 * n_barcodes), device=local)
    if args.sharded_index:
        from arcs_amd.dist import ShardedPairStep
        step = ShardedPairStep(index, reads, j, pair_ok=batch["pair_ok"], barcode_id=batch["barcode_id"], imap=imap)
    else:
        step = arcs_amd.PairStep(index, reads, j, pair_ok=batch["pair_ok"], barcode_id=batch["barcode_id"],
                                 imap=imap)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step.run()
    # one instrumented pass for the counters (outside the timed region)
    step.run(stats=None if args.sharded_index else stats, stored=stored)
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    t0 = time.perf_counter()
    for s in range(args.steps):
        step.run(map_events=ev[s])
    barrier()
    elapsed = time.perf_counter() - t0
    map_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    el = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    win = torch.tensor([float(windows)], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        if not args.sharded_index:      # sharded index: every rank worked on the same windows
            dist.all_reduce(win, op=dist.ReduceOp.SUM)
    elapsed_max, windows_all = float(el.item()), float(win.item())
    value = windows_all * args.steps / elapsed_max

    if rank == 0:
        st = dict(zip(("total_valid", "bad", "found", "recorded", "dups", "reads_pass", "reads_fail",
                       "windows"), stats.cpu().tolist()))
        assert st["windows"] <= windows
        achieved = windows * b_alg / (map_ms * 1e-3) / 1e9
        traffic = pmc_traffic(args)
        out = {
            "metric": METRIC, "value": value, "unit": "k-mers/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed_max / args.steps,
            "higher_is_better": True, "scaling": "strong" if args.sharded_index else "weak",
            "vs_baseline": None, "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": f"synthetic {args.draft_mbp:g} Mbp draft + {args.pairs} linked-read "
                                   f"pairs per GPU (R1 128 / R2 151 bp), k={k} j={j}",
                       "k": k, "j": j, "pairs_per_gpu": args.pairs, "windows_per_gpu": windows,
                       "index_keys": len(index),
                       "index_kind": "locality (text + minimizer table)" if index.kind == 1 else "hash table",
                       "index_bytes": index.device_bytes,
                       "parallelism": (f"index sharded x{world} (contigs dealt to the lightest shard), reads replicated, "
                                       "all-reduce(MAX) of votes") if args.sharded_index
                       else f"index replica x{world}, reads sharded"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": (traffic["hbm_bytes_per_launch"] / 1e9) if traffic else None,
                         "traffic_unit": "GB per launch (rocprofv3 PMC FETCH_SIZE + WRITE_SIZE)",
                         "traffic_source": traffic["source"] if traffic else None,
                         "alg_bytes_per_launch_GB": windows * b_alg / 1e9,
                         "kernel": "map_reads_b_kernel" if index.kind == 1 else "map_reads_kernel",
                         "kernel_ms": map_ms, "alg_bytes_per_window": b_alg},
            "counters": st, "stored_pairs": int(stored.item()),
        }
        if not args.no_cpu_baseline and world == 1:
            cb, (n, c, p, cst) = cpu_baseline(cs, batch, k, j, args.pairs, log)
            out["cpu_baseline"] = cb
            # parity guard on the sample: the GPU results of the same pairs must be identical
            got_c = step.conreci[: 2 * n].cpu().numpy()
            got_p = step.pair[:n].cpu().numpy()
            out["sample_parity"] = bool((got_c == c).all() and (got_p == p).all())
            assert out["sample_parity"], "GPU results differ from the CPU oracle on the sample"
            out["gpu_over_cpu"] = value / cb["value"]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
