#!/usr/bin/env python3
"""bench.py -- throughput of the ARKS read->contig k-mer mapping hot path on MI355X.

One "step" = one pass of the hot path (pair gate -> per-read k-mer window keys -> contig k-mer
index lookup -> per-read vote -> pair rule + (barcode, contig end) accumulation) over the whole
resident read set.  Workload at N=1: BASELINE.json configs[2] -- synthetic 3 Gbp draft + 500 M
linked-read pairs (R1 128 bp / R2 151 bp), k=60, j=0.55, everything resident in HBM (packed reads
~55 GB, seed index ~46 GB); the read set is mapped in launches of 250 M pairs.  The read set is a
function of the global pair number alone (generated in 40 blocks, block b from seed SEED+1+b), so
the same 500 M pairs are mapped whatever the number of ranks.  With N > 1 every rank holds a replica
of the index and the blocks are dealt to the ranks (strong scaling: fixed total work, no data-path
collective -- the path shards over reads, Arcs/Arcs.cpp:1169); --weak gives every rank its own
--pairs instead; --sharded-index shards the seed table over the ranks (BASELINE configs[3]).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--pairs P] [--draft-mbp M] [--weak]

`python bench.py --gpus N` starts its N ranks itself (re-executes under torch.distributed.run, one
process per GPU, backend nccl = RCCL); started under torch.distributed.run already (WORLD_SIZE set)
it is one of the ranks.  Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement").
"""
import argparse
import glob
import hashlib
import json
import os
import socket
import subprocess
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
STAT_NAMES = ("total_valid", "bad", "found", "recorded", "dups", "reads_pass", "reads_fail", "windows")


def alg_bytes_per_window(k, bases, windows):
    """packed read stream + one key compare + one int value per window"""
    return 2.0 * bases / (8.0 * windows) + (2 * k + 7) // 8 + 4


def kernel_build_id():
    """digest of the kernel sources the loaded library was built from; profiles/traffic_r*.json
    carries the id of the build its counters were taken on"""
    h = hashlib.sha256()
    for f in ("arks_device.hpp", "arks_kernels.hpp", "arks_map.hip"):
        h.update(open(os.path.join(ROOT, "arcs_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(workload):
    """HBM bytes (and L2 misses) per launch of the dominant kernel from the committed rocprofv3 PMC summaries
    (profiles/traffic_r*.json, written from separate --pmc passes by profiles/prof.sh).
    Returns (summary, build_match, scaled): the summary of this very workload taken on THIS kernel build when
    there is one; otherwise the newest summary of the workload with build_match = False (reported as such: the
    counters of another build are an estimate, never passed off as this build's).  A launch of another size over the
    same draft and k (a rank's share under --gpus N) takes the per-pair figures of the full-size summary times its
    pairs -- the kernel's traffic is per pair -- and says so (`scaled` = the pairs per launch it was scaled from);
    (None, False, None) when there is nothing to go by."""
    bid = kernel_build_id()
    exact, stale, other = None, None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_r*.json"))):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        w = d.get("workload", {})
        if w == workload:
            if d.get("kernel_build_id") == bid:
                exact = d
            else:
                stale = d
        elif {k: v for k, v in w.items() if k != "pairs_per_launch"} == \
                {k: v for k, v in workload.items() if k != "pairs_per_launch"} and w.get("pairs_per_launch"):
            if other is None or d.get("kernel_build_id") == bid or other.get("kernel_build_id") != bid:
                other = d
    if exact:
        return exact, True, None
    if stale:
        return stale, False, None
    if other:
        f = workload["pairs_per_launch"] / other["workload"]["pairs_per_launch"]
        d = dict(other)
        d["hbm_bytes_per_launch"] = other["hbm_bytes_per_launch"] * f
        if other.get("hbm_bytes_per_launch_uncalibrated"):
            d["hbm_bytes_per_launch_uncalibrated"] = other["hbm_bytes_per_launch_uncalibrated"] * f
        if other.get("TCC_MISS_per_launch"):
            d["TCC_MISS_per_launch"] = other["TCC_MISS_per_launch"] * f
        return d, other.get("kernel_build_id") == bid, other["workload"]["pairs_per_launch"]
    return None, False, None


def traffic_model(k, index, pairs_per_launch, read_lens, hbm_bytes_per_launch, misses_per_launch):
    """What THIS kernel design must move per read pair, beside what it does move (SURVEY 8(d)'s
    19.43 B per window describes one key compare per window, which the locality index does not do; `frac` alone rises
    when bytes are wasted).  The minimum of the design:
      * the read stream: the packed words of the pair, word offsets, lengths, the eval byte -- N masks are not fetched
        for tiles of ACGT-only reads (what profiles/r09_traffic_classes.json measured as 98.0 B on the 128 + 151 pair);
      * one seed-table probe per seed: ceil(windows / (k - m + 1)) seeds per read, one aligned group of four 8-byte
        entries = 32 B each -- the fabric fetches a 64-byte sector for it, which is the over-fetch `efficiency` shows
        (`at_64B_sectors`: the same minimum with the probe priced at the sector the memory system moves);
      * the text records along the read's diagonal, for the reads that have one: taken at the true bytes the calibration
        passes measured for this kind of read set (a class figure of the workload, not of the kernel build).
    efficiency = minimum / moved: it FALLS when the kernel wastes bytes (a probe made twice, a sector half used), unlike
    `frac`.  min_accesses: one per probe, one per 64-byte sector of stream and of text records."""
    m = 21 if k >= 24 else 17
    w = k - m + 1
    seeds = sum(-(-max(L - k + 1, 0) // w) for L in read_lens)
    stream = sum(8 * (-(-L // 32)) + 8 + 4 + 1 for L in read_lens)          # words + word_off + len + eval
    cls_path = os.path.join(ROOT, "profiles", "r09_traffic_classes.json")
    try:
        text = json.load(open(cls_path))["per_pair_bytes_corrected"]["text_records"]
    except (OSError, KeyError, ValueError):
        text = None
    if index.kind != 2 or text is None:
        return None
    mn32 = stream + 32.0 * seeds + text
    mn64 = stream + 64.0 * seeds + text
    moved = hbm_bytes_per_launch / pairs_per_launch if hbm_bytes_per_launch else None
    acc = seeds + stream / 64.0 + text / 64.0
    return {"unit": "bytes per read pair",
            "min_bytes": {"read_stream": stream, "probes": 32.0 * seeds, "text_records": text, "total": mn32},
            "min_bytes_at_64B_sectors": mn64,
            "seeds_per_pair": seeds,
            "moved_bytes": moved,
            "efficiency": (mn32 / moved) if moved else None,
            "efficiency_at_64B_sectors": (mn64 / moved) if moved else None,
            "min_accesses": acc,
            "l2_misses": (misses_per_launch / pairs_per_launch) if misses_per_launch else None,
            "access_efficiency": (acc * pairs_per_launch / misses_per_launch) if misses_per_launch else None,
            "text_records_source": "profiles/r09_traffic_classes.json per_pair_bytes_corrected"}


# measured ceiling of independent random HBM accesses on this part (profiles/r03_gather_tlb.txt: 3.8e10 /s over a
# 16-64 GiB table whatever the allocation, 4.8e10 /s with the probes of a launch confined to a 1 GiB slice)
RANDOM_ACCESS_CEILING = 3.8e10


# ---- host topology (for the CPU baseline) --------------------------------------------------------

def cpu_topology():
    """(model name, {socket: [one logical cpu per physical core]}) restricted to the cpus this
    process may run on"""
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    allowed = sorted(os.sched_getaffinity(0))
    sockets = {}
    seen = set()
    for c in allowed:
        base = f"/sys/devices/system/cpu/cpu{c}/topology/"
        try:
            pkg = int(open(base + "physical_package_id").read())
            core = int(open(base + "core_id").read())
        except (OSError, ValueError):
            pkg, core = 0, c
        if (pkg, core) in seen:
            continue
        seen.add((pkg, core))
        sockets.setdefault(pkg, []).append(c)
    return model, sockets


def cpu_quota():
    """CPUs' worth of time the container may use per period (cgroup v2 cpu.max, v1 cfs quota), None = no limit.
    The GPU boxes of this pool show 256 logical CPUs and allow 16 (cpu.max 1600000 100000,
    profiles/r04n_cpu_diag.txt): more busy threads than that are throttled, not run."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


# ---- workload ------------------------------------------------------------------------------------

N_BLOCKS = 40          # the read set is generated in this many blocks (500 M pairs: 12.5 M each)


def read_blocks(pairs):
    """[(first pair, pairs)] of the generation blocks of a read set of `pairs` pairs"""
    nb = max(1, min(N_BLOCKS, pairs))
    return [(b * pairs // nb, (b + 1) * pairs // nb - b * pairs // nb) for b in range(nb)]


def blocks_of_rank(n_blocks, rank, world):
    """contiguous share of the blocks (strong scaling): [lo, hi)"""
    return rank * n_blocks // world, (rank + 1) * n_blocks // world


class Workload:
    """a draft, its index, and a resident read set cut into launches.  The read set is made of generation
    blocks: block b of the set numbered `set_id` comes from seed SEED + 1 + set_id * N_BLOCKS + b and holds
    the pairs [first, first + n) of that set, so a rank that is given blocks [lo, hi) maps exactly the pairs
    a single rank would map there."""

    def __init__(self, draft_mbp, pairs, chunk, k, j, dev, local, log, blocks=None, set_id=0, want_stats=True,
                 keep_draft=False, repeats=False):
        self.k, self.j = k, j
        t0 = time.time()
        self.dup_events = []
        self.repeat_sites = []
        contigs = synth.make_draft(int(draft_mbp * 1e6), seed=synth.SEED, dup_events=self.dup_events,
                                   repeats=repeats, repeat_sites=self.repeat_sites)
        self.n_contigs = len(contigs)
        ends = []
        for c in contigs:
            cut = arcs_amd.end_cutoff(len(c))
            if cut is None:
                continue
            ends.append(c[:cut].tobytes())
            ends.append(c[len(c) - cut:].tobytes())
        log(f"draft: {len(contigs)} contigs, {sum(map(len, contigs))} bp, {len(ends)} ends in {time.time() - t0:.1f}s")
        t0 = time.time()
        self.index = arcs_amd.ArksIndex.build(ends, k, device=local, want_stats=want_stats)
        self.index_build_s = time.time() - t0
        del ends
        log(f"index: {len(self.index)} keys, {self.index.device_bytes / 2**30:.2f} GiB, built in "
            f"{time.time() - t0:.1f}s {self.index.build_stats}")
        t0 = time.time()
        self.genome = torch.from_numpy(np.concatenate(contigs)).to(dev)
        self.contigs = contigs if keep_draft else None
        all_blocks = read_blocks(pairs)
        lo, hi = blocks if blocks is not None else (0, len(all_blocks))
        self.windows_all = 0
        self.bases = 0
        self.pairs = 0
        launches, cur, cur_pairs = [], [], 0
        for b in range(lo, hi):
            first, n = all_blocks[b]
            batch = synth.make_read_pairs(self.genome, n, seed=synth.SEED + 1 + set_id * N_BLOCKS + b, device=dev)
            reads = arcs_amd.PackedReads.from_arrays_device(batch["ascii"], batch["offsets"], batch["lens"],
                                                            device=local)
            # one barcode per 80 pairs of the set (two molecules of 40 pairs), numbered over the whole set
            bid = ((torch.arange(n, device=dev, dtype=torch.int64) + (set_id * pairs + first)) // 80).to(torch.int32)
            self.windows_all += reads.windows(k)
            self.bases += int(batch["lens"].to(torch.int64).sum().item())
            cur.append((reads, batch["pair_ok"], bid))
            cur_pairs += n
            self.pairs += n
            del batch
            if cur_pairs >= chunk or b == hi - 1:
                launches.append((arcs_amd.PackedReads.concat([c[0] for c in cur]),
                                 torch.cat([c[1] for c in cur]), torch.cat([c[2] for c in cur])))
                cur, cur_pairs = [], 0
        self.imap = arcs_amd.ImapAccumulator(max(1 << 16, 8 * (self.pairs // 80 + 1)), device=local)   # it grows
        self.steps = [arcs_amd.PairStep(self.index, r, j, pair_ok=ok, barcode_id=b, imap=self.imap)
                      for r, ok, b in launches]
        self.pairs_per_launch = max([s.n_pairs for s in self.steps], default=0)
        log(f"reads: {self.pairs} pairs (blocks {lo}..{hi - 1} of {len(all_blocks)}) in {len(self.steps)} launches, "
            f"{self.windows_all} windows, resident in {time.time() - t0:.1f}s")

    def run(self, stats=None, stored=None, events=None):
        for i, s in enumerate(self.steps):
            s.run(stats=stats, stored=stored, map_events=None if events is None else events[i])

    def timed(self, steps, warmup, barrier):
        dev = self.genome.device
        stats = torch.zeros(8, dtype=torch.int64, device=dev)
        stored = torch.zeros(1, dtype=torch.int64, device=dev)
        for _ in range(warmup):
            self.run()
        self.run(stats=stats, stored=stored)          # the counters, outside the timed region
        barrier()
        ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
               for _ in self.steps] for _ in range(steps)]
        t0 = time.perf_counter()
        for s in range(steps):
            self.run(events=ev[s])
        barrier()
        elapsed = time.perf_counter() - t0
        launch_ms = [a.elapsed_time(b) for row in ev for a, b in row]
        st = dict(zip(STAT_NAMES, stats.cpu().tolist()))
        # the timed passes run the kernels WITHOUT the -v counters (another instantiation, with shortcuts of its own); the
        # IndexMap accumulated over warmup + 1 + steps identical passes: every count must be that many times the
        # counters pass's -- i.e. a multiple -- and the counts must add up to passes x the stored pairs
        passes = warmup + 1 + steps
        t = self.imap.triples()
        self.timed_path_parity = bool((t[:, 2] % passes == 0).all() and int(t[:, 2].sum()) == passes * int(stored.item()))
        return elapsed, launch_ms, st, int(stored.item())


def sub_draft_oracle(wl, log, sub_mbp):
    """(oracle index, bases of the sub-draft): the ends of the contigs of the first `sub_mbp` of the draft plus the
    contigs they share copied segments with, plus the windows around every (AT)n stretch and every planted repeat
    copy of the WHOLE draft (their k-mers recur between sites), numbered as in the whole draft -- gives the whole
    index's answers for reads drawn from the sub-draft (tests/test_oracle_subdraft.py)."""
    from oracle import pyoracle as O
    O.build_oracle()
    contigs = wl.contigs
    acc, n_first = 0, 0
    while n_first < len(contigs) and acc < sub_mbp * 1e6:
        acc += len(contigs[n_first])
        n_first += 1
    members = set(synth.closed_contig_set(n_first, wl.dup_events))
    t0 = time.time()
    runs = synth.alternating_at_runs(wl.genome, run=12)
    n_at = len(runs)
    if wl.repeat_sites:
        runs = np.concatenate([runs, synth.sites_to_runs(contigs, wl.repeat_sites)])
    ox = O.sub_draft_index(wl.k, contigs, members, site_runs=runs)
    log(f"cpu oracle index: ends of {len(members)} contigs + the windows around {n_at} (AT)n stretches and "
        f"{len(wl.repeat_sites)} repeat copies, {len(ox)} keys in {time.time() - t0:.1f}s")
    return ox, acc


def sample_against_oracle(wl, ox, acc, n_pairs, seed, threads, dev, local):
    """n_pairs read pairs of the workload's shape drawn from the sub-draft: the oracle on `threads` threads, the GPU
    against the WHOLE index -> (identical?, oracle seconds, oracle counters, pairs that reach into a microsatellite,
    host arrays for further oracle runs)"""
    batch = synth.make_read_pairs(wl.genome[:acc], n_pairs, seed=seed, device=dev)
    n_at = int(synth.pairs_touching_microsatellite(batch).sum().item())
    a = np.concatenate([batch["ascii"].cpu().numpy(), np.zeros(1, np.uint8)])
    lens = batch["lens"].cpu().numpy().astype(np.uint32)
    offs = batch["offsets"].cpu().numpy().astype(np.uint64)
    ok = batch["pair_ok"].cpu().numpy()
    t = time.time()
    c, p, st = ox.map_pairs(a, offs[: 2 * n_pairs], lens, wl.j, pair_ok=ok, threads=threads)
    dt = time.time() - t
    reads = arcs_amd.PackedReads.from_arrays_device(batch["ascii"], batch["offsets"], batch["lens"], device=local)
    got_c, got_p = arcs_amd.map_pairs_packed(wl.index, reads, wl.j, pair_ok=batch["pair_ok"])
    torch.cuda.synchronize(dev)
    same = bool((got_c.cpu().numpy() == c).all() and (got_p.cpu().numpy() == p).all())
    return same, dt, st, n_at, (a, offs, lens, ok)


def cpu_baseline(wl, dev, local, log, sub_mbp=50.0):
    """The CPU oracle (a literal port of the reference path: per-window O(k) re-encode, exact hash map,
    ordered histogram; OpenMP over pairs, Arcs.cpp:1169) timed at t=1 and on the physical cores of ONE
    socket ON THE SAME SAMPLE: reads of the workload's shape drawn from the first `sub_mbp` of the draft,
    as many pairs as one thread maps in ~15 s (sub_draft_oracle: the whole 1.4 G-key map does not fit a bench
    run).  Sample parity: the GPU maps a larger sample (4 M pairs, microsatellite reads included) against the
    WHOLE index and must agree read for read with the oracle.  Test infrastructure, never the product path."""
    k, j = wl.k, wl.j
    model, sockets = cpu_topology()
    sock = sorted(sockets)[0]
    cores = sockets[sock]
    quota = cpu_quota()
    if quota is not None and quota < len(cores):
        # the container's CPU quota is what the "socket" leg can really use: that many threads, one per core
        cores = cores[: max(1, int(quota))]
    saved = os.sched_getaffinity(0)
    os.sched_setaffinity(0, cores)             # OpenMP workers are created after this and inherit it
    try:
        ox, acc = sub_draft_oracle(wl, log, sub_mbp)
        n_max = 4_000_000
        parity, dtl, stl, n_at, (a_all, offs_all, lens_all, ok_all) = sample_against_oracle(
            wl, ox, acc, n_max, synth.SEED + 777, len(cores), dev, local)

        def run(n_pairs, threads):
            t = time.time()
            _, _, st = ox.map_pairs(a_all, offs_all[: 2 * n_pairs], lens_all[: 2 * n_pairs], j,
                                    pair_ok=ok_all[:n_pairs], threads=threads)
            return time.time() - t, st

        probe = 20000
        dt, _ = run(probe, 1)
        n_same = int(min(1_000_000, max(100_000, 15.0 * probe / max(dt, 1e-9))))
        dt1, st1 = run(n_same, 1)
        dts_same = min(run(n_same, len(cores))[0] for _ in range(3))
    finally:
        os.sched_setaffinity(0, saved)
    what = (f"{n_same} pairs ({st1['windows']} windows) of the workload's shape drawn from the first "
            f"{acc / 1e6:.0f} Mbp of the draft")
    out = {"value": st1["windows"] / dts_same, "unit": "k-mers/s", "cores": len(cores), "kind": "port",
           "cpu_model": model, "sockets_visible": len(sockets), "cpu_quota_cpus": quota,
           "physical_cores_visible": sum(len(v) for v in sockets.values()),
           "sample": f"{what}, {dts_same:.2f}s (best of 3), OpenMP {len(cores)} threads pinned one per physical core "
                     f"of socket {sock}" + (f" (the container's CPU quota is {quota:g} CPUs: as many threads as it can run)"
                                            if quota is not None else "") + f"; oracle index = the ends of those contigs ({len(ox)} keys, host RAM)",
           "t1": {"value": st1["windows"] / dt1, "unit": "k-mers/s", "cores": 1,
                  "sample": f"the same {what}, {dt1:.1f}s"},
           "scaling_over_t1": (st1["windows"] / dts_same) / (st1["windows"] / dt1),
           "large_sample": {"value": stl["windows"] / dtl, "unit": "k-mers/s", "cores": len(cores),
                            "sample": f"{n_max} pairs ({stl['windows']} windows), {dtl:.1f}s; {n_at} of them reach "
                                      f"into an (AT)n microsatellite; the GPU mapped the same reads against the "
                                      f"whole index: {'identical' if parity else 'DIFFERENT'}"}}
    return out, parity


def repeats_key(args, k, j, dev, local, log, barrier):
    """Secondary key `configs2_repeats`: the same 3 Gbp draft with human-like repeat families planted
    (synth.plant_repeats: 1e5 copies of a 300-bp element at 10-15 % divergence, 1e3 of a 6-kbp element, satellite
    arrays) -- reads that touch them carry heavy seeds and leave the hot kernel for the general ones -- and
    250 M read pairs in one launch (100 M until round 5, when the headline's launches were 100 M too); sample parity
    on 1 M pairs as for the headline."""
    pairs = min(args.pairs, 250_000_000)          # (one launch of the size the headline's are)
    wl = Workload(args.draft_mbp, pairs, pairs, k, j, dev, local, log, want_stats=False, keep_draft=True, repeats=True)
    el, l_ms, st, _ = wl.timed(max(2, args.steps // 2), 1, barrier)
    steps = max(2, args.steps // 2)
    q = arcs_amd.queue_counts(wl.index)
    ox, acc = sub_draft_oracle(wl, log, 30.0)
    same, dt, sto, n_at, _ = sample_against_oracle(wl, ox, acc, 1_000_000, synth.SEED + 778,
                                                   min(64, os.cpu_count() or 1), dev, local)
    return {"workload": f"synthetic {args.draft_mbp:g} Mbp draft with planted repeat families "
                        f"({len(wl.repeat_sites)} copies) + {pairs} linked-read pairs in one launch, k={k} j={j}",
            "value": st["windows"] * steps / el, "unit": "k-mers/s", "kernel_ms": float(np.mean(l_ms)),
            "kernel_ms_per_20M_pairs": float(np.mean(l_ms)) * 20_000_000 / max(1, pairs),
            "reads_left_to_general_kernels": {"medium": q[1], "slow": q[0], "of": 2 * pairs},
            "index_keys": len(wl.index), "index_bytes": wl.index.device_bytes,
            "sample_parity": same, "sample": f"1000000 pairs drawn from the first {acc / 1e6:.0f} Mbp, oracle "
                                             f"{dt:.1f}s; GPU against the whole index: "
                                             f"{'identical' if same else 'DIFFERENT'}"}


def human_like_key(args, k, j, dev, local, log, barrier):
    """Secondary key `configs2_human_like`: the 3 Gbp draft with a human-like repeat SPECTRUM --
    synth.plant_human_like: ~10 % SINE-like (families of 35 k copies of a 300-bp element, 5-20 % diverged) + ~10 %
    LINE-like (families of 2.5 k copies of a 6-kbp element, 3-15 %, most truncated) + satellite arrays -- and 250 M
    read pairs in one launch: k-mers/s, the reads the hot kernel leaves to the general kernels, the bytes of the exact
    table behind heavy seeds, the index build time.  Sample parity: the families have a FIXED size and their number
    scales with the draft, so a 100 Mbp draft of the same generator (one family of each class) shows a read the same
    multiplicities; 1 M pairs drawn from ALL of it, the GPU against the oracle over the WHOLE small draft (a sub-draft
    oracle of the 3 Gbp one would have to hold the windows around 1.1 M copies: 0.7 G keys)."""
    from oracle import pyoracle as O
    pairs = min(args.pairs, 250_000_000)          # (one launch of the size the headline's are)
    wl = Workload(args.draft_mbp, pairs, pairs, k, j, dev, local, log, want_stats=False, repeats="human")
    steps = max(2, args.steps // 2)
    el, l_ms, st, _ = wl.timed(steps, 1, barrier)
    q = arcs_amd.queue_counts(wl.index)
    fb_keys, fb_bytes = wl.index.fallback_size
    out = {"workload": f"synthetic {args.draft_mbp:g} Mbp draft with a human-like repeat spectrum ({len(wl.repeat_sites)} "
                       f"copies, {sum(e - b for _, b, e in wl.repeat_sites) / 1e6:.0f} Mbp = "
                       f"{100.0 * sum(e - b for _, b, e in wl.repeat_sites) / (args.draft_mbp * 1e6):.1f} % of the draft) + "
                       f"{pairs} linked-read pairs in one launch, k={k} j={j}",
           "value": st["windows"] * steps / el, "unit": "k-mers/s", "kernel_ms": float(np.mean(l_ms)),
           "kernel_ms_per_20M_pairs": float(np.mean(l_ms)) * 20_000_000 / max(1, pairs),
           "reads_left_to_general_kernels": {"medium": q[1], "slow": q[0], "of": 2 * pairs},
           "index_keys": len(wl.index), "index_bytes": wl.index.device_bytes, "index_build_s": wl.index_build_s,
           "fallback_keys": fb_keys, "fallback_bytes": fb_bytes, "timed_path_parity": wl.timed_path_parity}
    del wl
    torch.cuda.empty_cache()
    # the small draft of the same generator against the whole oracle
    small_mbp, n_pairs = 100.0, 1_000_000
    contigs = synth.make_draft(int(small_mbp * 1e6), seed=synth.SEED, repeats="human")
    ends = []
    for c in contigs:
        cut = arcs_amd.end_cutoff(len(c))
        if cut is not None:
            ends.append(c[:cut].tobytes())
            ends.append(c[len(c) - cut:].tobytes())
    ix = arcs_amd.ArksIndex.build(ends, k, device=local, want_stats=False)
    O.build_oracle()
    t0 = time.time()
    ox = O.OracleIndex(k).build(ends)
    t_ox = time.time() - t0
    del ends
    genome = torch.from_numpy(np.concatenate(contigs)).to(dev)
    batch = synth.make_read_pairs(genome, n_pairs, seed=synth.SEED + 780, device=dev)
    a = np.concatenate([batch["ascii"].cpu().numpy(), np.zeros(1, np.uint8)])
    lens = batch["lens"].cpu().numpy().astype(np.uint32)
    offs = batch["offsets"].cpu().numpy().astype(np.uint64)
    ok = batch["pair_ok"].cpu().numpy()
    t0 = time.time()
    c, p_, sto = ox.map_pairs(a, offs[: 2 * n_pairs], lens, j, pair_ok=ok, threads=min(64, os.cpu_count() or 1))
    dt = time.time() - t0
    reads = arcs_amd.PackedReads.from_arrays_device(batch["ascii"], batch["offsets"], batch["lens"], device=local)
    got_c, got_p = arcs_amd.map_pairs_packed(ix, reads, j, pair_ok=batch["pair_ok"])
    stc = torch.zeros(8, dtype=torch.int64, device=dev)
    got_c2, got_p2 = arcs_amd.map_pairs_packed(ix, reads, j, pair_ok=batch["pair_ok"], stats=stc)
    torch.cuda.synchronize(dev)
    same = bool((got_c.cpu().numpy() == c).all() and (got_p.cpu().numpy() == p_).all()
                and (got_c2.cpu().numpy() == c).all() and (got_p2.cpu().numpy() == p_).all()
                and dict(zip(STAT_NAMES, stc.cpu().tolist())) == {f: sto[f] for f in STAT_NAMES})
    qs = arcs_amd.queue_counts(ix)
    out["sample_parity"] = same
    out["sample"] = (f"{n_pairs} pairs drawn from a {small_mbp:g} Mbp draft of the same generator ({len(contigs)} contigs; "
                     f"one family of each class, the same copies per family), oracle over the WHOLE draft ({len(ox)} keys, "
                     f"built in {t_ox:.1f}s, mapped in {dt:.1f}s): per-read results, pair results (with and without "
                     f"counters) and the eight counters {'identical' if same else 'DIFFERENT'}; "
                     f"{int((c != 0).sum())} reads with a contig end; medium queue {qs[1]}, slow {qs[0]} of {2 * n_pairs}")
    return out


def end_to_end_files(wl, dev, log, sub_mbp=20.0, n_pairs=2_000_000, threads=16):
    """SURVEY 8(d), "additionally end-to-end from .fq.gz": ONE gzipped interleaved FASTQ of n_pairs read pairs drawn
    from the first sub_mbp of the draft (+ that sub-draft as FASTA and a barcode multiplicity file), mapped
      * by the CPU port from the file (oracle/arks_port_fastq.c: chromiumRead's loop, records read inside one
        critical section as in Arcs/Arcs.cpp:1185, mapped by the physical cores of one socket), and
      * by the product's front end `arcs --arks` (arcs_amd/bin/arcs: parallel gzip decode, parse, pack, H2D, the
        kernels, graph, output files) at -t `threads`,
    same files, same box.  Both must store the same number of read pairs."""
    import re
    import shutil
    import tempfile
    from oracle import pyoracle as O
    from arcs_amd import build as ab
    exe = ab.build_host()
    k, j = wl.k, wl.j
    contigs = wl.contigs
    acc, n_first = 0, 0
    while n_first < len(contigs) and acc < sub_mbp * 1e6:
        acc += len(contigs[n_first])
        n_first += 1
    members = synth.closed_contig_set(n_first, wl.dup_events)
    n_pairs -= n_pairs % 80
    tmp = tempfile.mkdtemp(prefix="arks_e2e_")
    try:
        t0 = time.time()
        with open(os.path.join(tmp, "draft.fa"), "wb") as f:
            for ci in members:
                f.write(b">%d\n" % (ci + 1))
                f.write(contigs[ci].tobytes())
                f.write(b"\n")
        batch = synth.make_read_pairs(wl.genome[:acc], n_pairs, seed=synth.SEED + 779, device=dev)
        text = synth.fastq_bytes(batch)
        fq = os.path.join(tmp, "reads.fq.gz")
        synth.write_gz_members(fq, text, threads=16, member_bytes=256 << 20)
        bid = batch["barcode_id"].cpu().numpy().astype(np.int64)
        with open(os.path.join(tmp, "mult.tsv"), "w") as f:
            for b in range(int(bid.max()) + 1):
                v, name = b, []
                for _ in range(16):
                    name.append("ACGT"[v % 4])
                    v //= 4
                f.write("".join(reversed(name)) + "-1\t160\n")
        windows = int(torch.clamp(batch["lens"].to(torch.int64) - (k - 1), min=0).sum().item())
        gz_mb, text_mb = os.path.getsize(fq) / 1e6, text.size / 1e6
        del batch, text
        log(f"end to end: {n_pairs} pairs as one .fq.gz ({gz_mb:.0f} MB, {text_mb:.0f} MB of text) + sub-draft of "
            f"{len(members)} contigs written in {time.time() - t0:.1f}s")
        # the CPU port
        model, sockets = cpu_topology()
        cores = sockets[sorted(sockets)[0]]
        quota = cpu_quota()
        if quota is not None and quota < len(cores):
            cores = cores[: max(1, int(quota))]        # (what the container may really run at once)
        saved = os.sched_getaffinity(0)
        os.sched_setaffinity(0, cores)
        try:
            ox = O.sub_draft_index(k, [contigs[ci] for ci in members], range(len(members)))
            t0 = time.time()
            got_pairs, cpu_stored, st = O.map_fastq_gz(ox, fq, j, threads=len(cores))
            cpu_s = time.time() - t0
        finally:
            os.sched_setaffinity(0, saved)
        assert got_pairs == n_pairs, (got_pairs, n_pairs)
        # the product's front end
        t0 = time.time()
        res = subprocess.run([exe, "--arks", "-v", "-f", os.path.join(tmp, "draft.fa"), "-u", os.path.join(tmp, "mult.tsv"),
                              "-k", str(k), "-j", str(j), "-c", "5", "-m", "50-10000", "-e", "30000", "-z", "500",
                              "-t", str(threads), "-b", os.path.join(tmp, "out"), fq],
                             capture_output=True, text=True, env=dict(os.environ, ARKS_TIMING="1"))
        cli_s = time.time() - t0
        assert res.returncode == 0, res.stderr[-2000:]
        m = re.search(r"Stored read pairs: (\d+)", res.stdout)
        cli_stored = int(m.group(1)) if m else -1
        rd = [ln for ln in res.stderr.splitlines() if "read files" in ln]
        read_ms = float(rd[0].split(":")[1].split()[0]) if rd else None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return {"input": f"one gzipped interleaved FASTQ, {n_pairs} pairs ({gz_mb:.0f} MB; {text_mb:.0f} MB of text; gzip "
                     f"members of 256 MB of text), reads drawn from a {acc / 1e6:.0f} Mbp sub-draft ({len(members)} contigs as FASTA)",
            "windows": windows,
            "cpu_port": {"value": st["windows"] / cpu_s, "unit": "k-mers/s", "pairs_per_s": n_pairs / cpu_s,
                         "seconds": cpu_s, "cores": len(cores), "stored_pairs": cpu_stored,
                         "what": "oracle/arks_port_fastq.c: records read inside one critical section (gzgets), mapped "
                                 "by OpenMP threads pinned to the physical cores of one socket; index prebuilt"},
            "gpu_cli": {"value": windows / cli_s, "unit": "k-mers/s", "pairs_per_s": n_pairs / cli_s, "seconds": cli_s,
                        "read_stage_ms": read_ms,
                        "read_stage_pairs_per_s": (n_pairs / (read_ms * 1e-3)) if read_ms else None,
                        "threads": threads, "stored_pairs": cli_stored,
                        "what": "arcs --arks -v, whole process: start-up, draft FASTA, index build on the device, the read "
                                "stage (parallel gzip decode, parse, pack, H2D, kernels), graph and output files"},
            "same_stored_pairs": cli_stored == cpu_stored}


def end_to_end_scale(wl, dev, local, log, port, n_files=8, pairs_per_file=4_000_000, threads=16):
    """`end_to_end.from_fq_gz_at_scale`: the product's front end on a run of a size where start-up
    no longer decides -- the WHOLE draft as FASTA (3 GB) and n_files gzipped interleaved FASTQ files of pairs_per_file
    pairs drawn from all of it (32 M pairs, ~20 GB of text), `arcs --arks -v -t 16` as a whole process: draft read,
    index build on the device (1.4 G keys), parallel gzip decode, parse, pack, H2D, kernels, graph, output files.  The
    CPU port cannot hold the whole draft's map in a bench run, so its side is the rate it reached on the small
    from_fq_gz leg (`port`: a sub-draft's map in host RAM, reads of that sub-draft, the same read shape) -- an
    extrapolation, labelled as one: per pair the port's work does not depend on the draft (one hash probe per
    window, out of cache in both).  Stored pairs are checked against the library path: the same reads mapped through
    the C ABI from device arrays against the resident whole index."""
    import re
    import shutil
    import tempfile
    from arcs_amd import build as ab
    exe = ab.build_host()
    k, j = wl.k, wl.j
    tmp = tempfile.mkdtemp(prefix="arks_e2e_scale_")
    free = shutil.disk_usage(tmp).free
    need = int(sum(len(c) for c in wl.contigs) * 1.02 + n_files * pairs_per_file * 130)
    scaled = None
    if free < 2 * need:            # (a small scratch disk: fewer pairs, and the record says so)
        scaled = max(1, int(n_files * (free / (2.0 * need))))
        log(f"end to end at scale: {free / 1e9:.1f} GB free in {tmp}, {need / 1e9:.1f} GB wanted: {scaled} files instead of {n_files}")
        n_files = scaled
    pairs_per_file -= pairs_per_file % 80
    try:
        t0 = time.time()
        with open(os.path.join(tmp, "draft.fa"), "wb") as f:
            for ci, c in enumerate(wl.contigs):
                f.write(b">%d\n" % (ci + 1))
                f.write(c.tobytes())
                f.write(b"\n")
        t_fa = time.time() - t0
        t0 = time.time()
        files, windows, lib_stored, gz_b, text_b, max_bid = [], 0, 0, 0, 0, 0
        for fi in range(n_files):
            batch = synth.make_read_pairs(wl.genome, pairs_per_file, seed=synth.SEED + 900 + fi, device=dev)
            text = synth.fastq_bytes(batch, first_pair=fi * pairs_per_file)
            fq = os.path.join(tmp, f"reads{fi}.fq.gz")
            synth.write_gz_members(fq, text, threads=16, member_bytes=256 << 20)
            files.append(fq)
            gz_b += os.path.getsize(fq)
            text_b += text.size
            windows += int(torch.clamp(batch["lens"].to(torch.int64) - (k - 1), min=0).sum().item())
            max_bid = max(max_bid, int(batch["barcode_id"].max().item()))
            # the library path on the same reads (device arrays, the resident whole index): what the CLI must store
            reads = arcs_amd.PackedReads.from_arrays_device(batch["ascii"], batch["offsets"], batch["lens"], device=local)
            _, pr = arcs_amd.map_pairs_packed(wl.index, reads, j, pair_ok=batch["pair_ok"])
            lib_stored += int(((pr != 0) & (batch["pair_ok"] != 0)).sum().item())
            del batch, text, reads, pr
        with open(os.path.join(tmp, "mult.tsv"), "w") as f:
            for b in range(max_bid + 1):
                v, name = b, []
                for _ in range(16):
                    name.append("ACGT"[v % 4])
                    v //= 4
                f.write("".join(reversed(name)) + "-1\t160\n")
        n_pairs = n_files * pairs_per_file
        log(f"end to end at scale: draft FASTA in {t_fa:.1f}s, {n_pairs} pairs as {n_files} .fq.gz ({gz_b / 1e9:.1f} GB, "
            f"{text_b / 1e9:.1f} GB of text) in {time.time() - t0:.1f}s")
        torch.cuda.empty_cache()
        t0 = time.time()
        res = subprocess.run([exe, "--arks", "-v", "-f", os.path.join(tmp, "draft.fa"), "-u", os.path.join(tmp, "mult.tsv"),
                              "-k", str(k), "-j", str(j), "-c", "5", "-m", "50-10000", "-e", "30000", "-z", "500",
                              "-t", str(threads), "-b", os.path.join(tmp, "out")] + files,
                             capture_output=True, text=True, env=dict(os.environ, ARKS_TIMING="1"))
        cli_s = time.time() - t0
        assert res.returncode == 0, res.stderr[-2000:]
        if os.environ.get("ARKS_BENCH_E2E_STDERR"):     # (profiling runs: the front end's own timing / ingest profile lines)
            open(os.environ["ARKS_BENCH_E2E_STDERR"], "w").write(res.stderr)
        m = re.findall(r"Stored read pairs: (\d+)", res.stdout)      # (per file: chromiumRead's locals, Arcs.cpp:1143-1146, 1322)
        cli_stored = sum(int(x) for x in m) if m else -1
        stages = {}
        for ln in res.stderr.splitlines():
            mm = re.match(r"\[timing\] +([^:]+): (\d+) ms", ln)
            if mm:
                stages[mm.group(1).strip()] = stages.get(mm.group(1).strip(), 0) + int(mm.group(2))
        rd = [v for kk, v in stages.items() if "read files" in kk]
        read_ms = float(rd[0]) if rd else None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    port_pairs_s = port["pairs_per_s"]
    whole = n_pairs / cli_s
    read_stage = (n_pairs / (read_ms * 1e-3)) if read_ms else None
    return {"input": f"{n_files} gzipped interleaved FASTQ files of {pairs_per_file} pairs ({gz_b / 1e9:.1f} GB; {text_b / 1e9:.1f} GB of "
                     f"text), reads drawn from the whole {sum(len(c) for c in wl.contigs) / 1e6:.0f} Mbp draft (FASTA, {len(wl.contigs)} contigs)"
                     + (f" [scaled down from 8 files: scratch disk]" if scaled else ""),
            "pairs": n_pairs, "windows": windows,
            "gpu_cli": {"seconds": cli_s, "pairs_per_s": whole, "value": windows / cli_s, "unit": "k-mers/s",
                        "read_stage_ms": read_ms, "read_stage_pairs_per_s": read_stage,
                        "read_stage_kmers_per_s": (windows / (read_ms * 1e-3)) if read_ms else None,
                        "stage_ms": stages, "threads": threads, "stored_pairs": cli_stored,
                        "what": "arcs --arks -v, whole process: start-up, the 3 Gbp draft FASTA, index build on the device, the read "
                                "stage (one decoding thread per .gz file, parse, pack, H2D, kernels), graph and output files"},
            "library_stored_pairs": lib_stored, "same_stored_pairs_as_library": cli_stored == lib_stored,
            "cpu_port_pairs_per_s": port_pairs_s,
            "cpu_port_note": "the port's rate on the from_fq_gz leg (a sub-draft's map, reads of that sub-draft, the same read "
                             "shape, the same host cores): an EXTRAPOLATION to this input -- the whole draft's map does not fit a bench run",
            "cli_over_port_whole_process": whole / port_pairs_s,
            "cli_read_stage_over_port": (read_stage / port_pairs_s) if read_stage else None,
            "bound": (f"whole process {cli_s:.1f}s, of which the read stage {read_ms / 1e3:.1f}s: the rest is start-up (draft FASTA, "
                      f"index build, graph, outputs); the read stage runs on the container's CPU quota "
                      f"({cpu_quota()} CPUs): gzip inflate is what its threads do") if read_ms else None}


def end_to_end(wl, dev, local, n_batches=8, pairs=2_000_000):
    """packed batches start in pinned HOST memory; per batch H2D (codes, N mask, offsets,
    lengths, class, pair_ok, barcode ids) -> gate / map / pair rule -> D2H (pair results), double-buffered
    on two streams.  The PCIe-inclusive rate, never `value`."""
    host = []
    windows = 0
    for b in range(2):
        s = wl.steps[b % len(wl.steps)]
        r = s.reads
        n = min(pairs, s.n_pairs)
        w = int(r.word_off[2 * n].item())
        host.append({nm: t.cpu().pin_memory() for nm, t in
                     dict(codes=r.codes[: w + 4], nmask=r.nmask[: w + 4], woff=r.word_off[: 2 * n + 1],
                          lens=r.lens[: 2 * n], cls=r.read_class[: 2 * n], ok=s.pair_ok[:n],
                          bid=s.barcode_id[:n]).items()})
        windows = int(torch.clamp(r.lens[: 2 * n].to(torch.int64) - (wl.k - 1), min=0).sum().item())
    pairs = host[0]["ok"].numel()
    if host[1]["ok"].numel() != pairs:
        host[1] = host[0]
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    dbuf = [{nm: torch.empty_like(t, device=dev) for nm, t in host[0].items()} for _ in range(2)]
    outs = [torch.empty(pairs, dtype=torch.int32).pin_memory() for _ in range(2)]
    steps = []
    for s in range(2):
        d = dbuf[s]
        reads = arcs_amd.PackedReads(d["codes"], d["nmask"], d["woff"], d["lens"], d["cls"], local)
        steps.append(arcs_amd.PairStep(wl.index, reads, wl.j, pair_ok=d["ok"], barcode_id=d["bid"]))
    bytes_in = sum(t.numel() * t.element_size() for t in host[0].values())

    def run(nb):
        for b in range(nb):
            s = b & 1
            with torch.cuda.stream(streams[s]):
                for nm, t in host[s].items():
                    dbuf[s][nm].copy_(t, non_blocking=True)
                steps[s].run()          # (an index keeps a set of work queues per stream: no ordering needed)
                outs[s].copy_(steps[s].pair[:pairs], non_blocking=True)
        torch.cuda.synchronize(dev)

    run(2)
    t0 = time.perf_counter()
    run(n_batches)
    dt = time.perf_counter() - t0
    return {"value": n_batches * windows / dt, "unit": "k-mers/s",
            "h2d_GBps": n_batches * bytes_in / dt / 1e9, "ms_per_batch": 1e3 * dt / n_batches,
            "what": f"{n_batches} packed batches of {pairs} pairs from pinned host memory: H2D {bytes_in / 1e6:.0f} MB "
                    f"+ gate/map/pairs + D2H of the pair results per batch, two streams"}


def free_port():
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launch_ranks(n):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks (one process per GPU, or
    N processes sharing the visible GPUs with ARKS_BENCH_BACKEND=gloo) and pass their output through"""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=500_000_000, help="read pairs in total (per GPU with --weak)")
    ap.add_argument("--chunk", type=int, default=250_000_000,
                    help="read pairs per launch (250 M: 2 launches per pass; the general kernels behind the hot one are a "
                         "latency-bound tail of every launch: 100 M -- rounds 2-4's default -- costs 2.6 %% more per pair, "
                         "20 M another 3-4 %%; ONE launch of 500 M is slower again, profiles/r09h_chunk.txt)")
    ap.add_argument("--draft-mbp", type=float, default=3000.0)
    ap.add_argument("--k", type=int, default=60)
    ap.add_argument("--j", type=float, default=0.55)
    ap.add_argument("--weak", action="store_true",
                    help="N > 1: every rank maps its own --pairs read pairs (per-GPU work fixed) instead of a share "
                         "of the one read set (the default: strong scaling on the fixed workload)")
    ap.add_argument("--strong", action="store_true", help="(the default; kept for older command lines)")
    ap.add_argument("--human-like", action="store_true",
                    help="only the configs2_human_like key (the draft with a human-like repeat spectrum): for profiles")
    ap.add_argument("--repeats", action="store_true",
                    help="the headline workload on the draft with planted repeat families (what the configs2_repeats "
                         "key runs at 100 M pairs): for profiles")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the configs[1] line and the end-to-end figure")
    ap.add_argument("--shards", type=int, default=8,
                    help="--sharded-index with one process: shards (local ranks) the seed table is cut into")
    ap.add_argument("--sharded-index", action="store_true",
                    help="BASELINE configs[3]: the index's seed table sharded over the ranks by a hash prefix of the "
                         "m-mer, the --pairs read pairs split over the ranks, every seed routed to its owner and "
                         "answered there (all-to-all); default: index replicas, reads sharded, no data-path collective")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # one process per GPU; ARKS_BENCH_BACKEND=gloo lets several ranks share one GPU (tests of the multi-rank
    # control flow on a single-GPU box; the driver's runs use nccl = RCCL)
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

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    assert arcs_amd.device_count() >= 1, "libarks_hip sees no gfx950 device (no CPU fallback)"
    k, j = args.k, args.j
    if args.sharded_index:
        return sharded_index_bench(args, world, rank, local, dev, red_dev, log, barrier)

    if args.human_like:
        print(json.dumps({"configs2_human_like": human_like_key(args, k, j, dev, local, log, barrier)}), flush=True)
        return
    weak = args.weak and world > 1
    n_blocks = len(read_blocks(args.pairs))
    blocks = None if weak else blocks_of_rank(n_blocks, rank, world)
    cpu_leg = (not args.no_cpu_baseline) and world == 1
    wl = Workload(args.draft_mbp, args.pairs, args.chunk, k, j, dev, local, log, blocks=blocks,
                  set_id=rank if weak else 0, want_stats=(rank == 0), keep_draft=cpu_leg, repeats=args.repeats)
    elapsed, launch_ms, st, stored = wl.timed(args.steps, args.warmup, barrier)

    el = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    tot = torch.tensor([float(st[n]) for n in STAT_NAMES] + [float(stored), float(wl.pairs), float(wl.windows_all)],
                       dtype=torch.float64, device=red_dev)          # exact below 2^53
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    elapsed_max = float(el.item())
    tot = [int(x) for x in tot.tolist()]
    st_job = dict(zip(STAT_NAMES, tot[:8]))
    stored_job, pairs_job, windows_all_job = tot[8], tot[9], tot[10]
    # the unit (Arcs.cpp:959-962): windows of the reads that reach bestContig
    value = st_job["windows"] * args.steps / elapsed_max

    if rank == 0:
        assert st["windows"] <= wl.windows_all
        b_alg = alg_bytes_per_window(k, wl.bases, wl.windows_all)
        n_launch = len(wl.steps)
        kernel_ms = float(np.mean(launch_ms))                     # mean launch of the map stage (rank 0)
        win_per_launch = st["windows"] / n_launch
        alg_achieved = win_per_launch * b_alg / (kernel_ms * 1e-3) / 1e9
        workload = {"draft_mbp": args.draft_mbp, "pairs_per_launch": wl.pairs_per_launch, "k": k}
        if args.repeats:
            workload["repeats"] = True
        traffic, build_match, traffic_scaled = pmc_traffic(workload)
        hbm_gb = (traffic["hbm_bytes_per_launch"] / 1e9) if traffic else None
        achieved = (hbm_gb / (kernel_ms * 1e-3)) if traffic else None
        misses = traffic.get("TCC_MISS_per_launch") if traffic else None
        model = traffic_model(k, wl.index, wl.pairs_per_launch, synth.READ_LENS,
                              traffic["hbm_bytes_per_launch"] if traffic else None, misses)
        scaling = "weak" if args.weak else "strong"          # the default: one fixed read set whatever N
        out = {
            "metric": METRIC, "value": value, "unit": "k-mers/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed_max / args.steps,
            "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": f"synthetic {args.draft_mbp:g} Mbp draft + {args.pairs} linked-read pairs "
                                   f"{'per GPU' if weak else 'in total'} (R1 128 / R2 151 bp), "
                                   f"k={k} j={j}, resident in HBM, mapped in launches of {wl.pairs_per_launch} pairs"
                                   + (" [BASELINE configs[2]]" if args.draft_mbp == 3000 and args.pairs == 500_000_000 else ""),
                       "k": k, "j": j, "pairs_job": pairs_job, "pairs_rank0": wl.pairs, "launches_per_step": n_launch,
                       "windows_job": st_job["windows"], "windows_incl_gated_reads": windows_all_job,
                       "index_keys": len(wl.index),
                       "index_kind": {0: "hash table", 1: "locality (text + minimizer table)",
                                      2: "locality (text + seed table: every m-mer position)"}[wl.index.kind],
                       "index_bytes": wl.index.device_bytes,
                       "parallelism": f"index replica x{world}, reads sharded"
                                      + ("" if world == 1 else (", every rank its own read set" if weak else
                                                                ", the one read set dealt to the ranks in blocks"))},
            # `frac` = what the dominant kernel moves through HBM per second (rocprofv3 FETCH_SIZE + WRITE_SIZE per
            # launch / its mean duration by HIP events in THIS run) over the 8 TB/s peak -- a fraction by construction.
            # The path's accesses are random 32-byte probes of a 46 GB table and 80-110-byte runs of text records,
            # so what bounds it is the rate of independent HBM accesses (`random_access`), not the byte rate.
            # `alg_*` = SURVEY 8(d)'s algorithmic figure (a 15-byte key compare + a value per window): the locality
            # index does not move those bytes, so alg_frac may pass 1 -- it compares the path with "one key compare
            # per window at HBM speed" and is NOT a roofline fraction.
            "roofline": {"bound": "hbm",
                         "bound_detail": "HBM random-access rate (32-byte seed-table probes + short runs of 16-byte "
                                         "text records), not byte bandwidth",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved is not None else None,
                         # utilisation (above) says how busy the memory system is; efficiency says how much of what it
                         # moves the design needs: minimum bytes of THIS kernel design per pair / bytes moved per pair
                         "efficiency": (model or {}).get("efficiency"),
                         "model": model,
                         "traffic": hbm_gb,
                         "traffic_unit": "GB per launch (rocprofv3 PMC, mean over the profiled dispatches: FETCH_SIZE x the "
                                         "calibrated read factor + WRITE_SIZE; FETCH_SIZE counts a 128-byte request as 64)",
                         "traffic_calibration": traffic.get("traffic_calibration") if traffic else None,
                         "traffic_uncalibrated": (traffic.get("hbm_bytes_per_launch_uncalibrated", 0) / 1e9 or None)
                         if traffic else None,
                         "traffic_source": traffic["source"] if traffic else None,
                         "traffic_build_match": build_match,
                         "traffic_scaled_from_pairs_per_launch": traffic_scaled,
                         # the kernel's L2 misses per second (random probes, short runs of text records, and the
                         # sequential read stream together) beside what the part sustains with random 32-byte reads
                         # alone: a ratio near 1 (the sequential share can take it past 1) says the memory system's
                         # request rate, not its byte rate, is what is used up
                         "random_access": ({"l2_misses_per_s": misses / (kernel_ms * 1e-3),
                                            "random_gather_ceiling_per_s": RANDOM_ACCESS_CEILING,
                                            "ratio": misses / (kernel_ms * 1e-3) / RANDOM_ACCESS_CEILING,
                                            "source": "TCC_MISS of the counter passes; profiles/r03_gather_tlb.txt "
                                                      "(3.8e10 random 32-byte reads/s over a 16-64 GiB table)"}
                                           if misses else None),
                         "kernel_build_id": kernel_build_id(),
                         "alg_achieved": alg_achieved, "alg_frac": alg_achieved / HBM_PEAK_GBS,
                         "alg_bytes_per_launch_GB": win_per_launch * b_alg / 1e9,
                         "alg_bytes_per_window": b_alg,
                         "kernel": {0: "map_reads_kernel", 1: "map_reads_b_kernel", 2: "map_reads_s_kernel"}[wl.index.kind],
                         "kernel_ms": kernel_ms, "launches_timed": len(launch_ms),
                         "kernel_ms_per_20M_pairs": kernel_ms * 20_000_000 / max(1, wl.pairs_per_launch)},
            "counters": st_job, "stored_pairs": stored_job,
            "timed_path_parity": wl.timed_path_parity,   # (rank 0's: the passes without counters stored what the pass with them did)
        }
        if cpu_leg:
            cb, parity = cpu_baseline(wl, dev, local, log)
            out["cpu_baseline"] = cb
            out["sample_parity"] = parity
            out["gpu_over_cpu_socket"] = value / cb["value"]
            out["gpu_over_cpu_t1"] = value / cb["t1"]["value"]
        if world == 1 and not args.no_extras:
            out["end_to_end"] = end_to_end(wl, dev, local)
            if cpu_leg:
                out["end_to_end"]["from_fq_gz"] = end_to_end_files(wl, dev, log)
                if args.draft_mbp >= 1000 or os.environ.get("ARKS_BENCH_E2E_SCALE"):
                    out["end_to_end"]["from_fq_gz_at_scale"] = end_to_end_scale(
                        wl, dev, local, log, out["end_to_end"]["from_fq_gz"]["cpu_port"],
                        n_files=int(os.environ.get("ARKS_BENCH_E2E_FILES", 8)),
                        pairs_per_file=int(os.environ.get("ARKS_BENCH_E2E_PAIRS_PER_FILE", 4_000_000)))
            del wl
            torch.cuda.empty_cache()
            # BASELINE configs[1] (round 1's headline), same build, same box
            c2 = Workload(50.0, 20_000_000, 20_000_000, k, j, dev, local, log, want_stats=False)
            e2, l2, s2, _ = c2.timed(args.steps, args.warmup, barrier)
            b2 = alg_bytes_per_window(k, c2.bases, c2.windows_all)
            del c2
            torch.cuda.empty_cache()
            out["configs2_repeats"] = repeats_key(args, k, j, dev, local, log, barrier)
            torch.cuda.empty_cache()
            out["configs2_human_like"] = human_like_key(args, k, j, dev, local, log, barrier)
            out["configs1"] = {"workload": "synthetic 50 Mbp draft + 20000000 linked-read pairs, k=60 j=0.55",
                               "value": s2["windows"] * args.steps / e2, "unit": "k-mers/s",
                               "kernel_ms": float(np.mean(l2)),
                               "alg_frac": s2["windows"] * b2 / (float(np.mean(l2)) * 1e-3) / 1e9 / HBM_PEAK_GBS}
        print(json.dumps(out), flush=True)
        if cpu_leg:
            assert out["sample_parity"], "GPU results differ from the CPU oracle on the sample"
            assert out["timed_path_parity"], "the timed passes (kernels without counters) stored other pairs than the counters pass"
        if "from_fq_gz" in out.get("end_to_end", {}):
            e2e = out["end_to_end"]["from_fq_gz"]
            assert e2e["gpu_cli"]["stored_pairs"] < 0 or e2e["same_stored_pairs"], \
                "CLI and CPU port store different numbers of pairs"
        if "from_fq_gz_at_scale" in out.get("end_to_end", {}):
            e2s = out["end_to_end"]["from_fq_gz_at_scale"]
            assert e2s["same_stored_pairs_as_library"], "CLI at scale and the library path store different numbers of pairs"
        if "configs2_repeats" in out:
            assert out["configs2_repeats"]["sample_parity"], "repeat-rich draft: GPU results differ from the CPU oracle"
        if "configs2_human_like" in out:
            assert out["configs2_human_like"]["sample_parity"], "human-like draft: GPU results differ from the CPU oracle"
    if world > 1:
        dist.destroy_process_group()


def sharded_index_bench(args, world, rank, local, dev, red_dev, log, barrier):
    """BASELINE configs[3]: the seed table of the index sharded by a hash prefix of the m-mer
    (arks_index_build_seed_shard), the --pairs read pairs dealt to the ranks in blocks (strong scaling), every seed
    routed to its owner and answered there -- arks_exchange (include/arks_hip.h): seeds bucketed by owner on the
    device, ncclSend / ncclRecv groups over RCCL, owner-side probe, answers back, map_reads_s_kernel<REMOTE>;
    DESIGN.md 6.  One step = gate -> exchanged map -> pair rule over a rank's reads, in launches of --chunk pairs.
    With ONE process (N = 1) the table is cut into --shards shards all the same (default 8), every shard a local
    rank with a host thread of its own on the one device: the data path of 8 ranks, timed on one GPU."""
    import threading
    import torch.distributed as dist
    from arcs_amd import dist as adist
    k, j = args.k, args.j
    n_local = max(1, args.shards) if world == 1 else 1
    if n_local > 1:
        # the exchange buffers of all local ranks live on the one device (~155 B per pair and rank for two batches in flight)
        args.chunk = min(args.chunk, int(os.environ.get("ARKS_BENCH_SHARDED_CHUNK", 12_500_000)))
    n_ranks = world * n_local
    contigs = synth.make_draft(int(args.draft_mbp * 1e6), seed=synth.SEED)
    ends = []
    for c in contigs:
        cut = arcs_amd.end_cutoff(len(c))
        if cut is not None:
            ends.append(c[:cut].tobytes())
            ends.append(c[len(c) - cut:].tobytes())
    t0 = time.time()
    my_ranks = [rank * n_local + i for i in range(n_local)]
    shards = [arcs_amd.ArksIndex.build_seed_shard(ends, k, r, n_ranks, device=local) for r in my_ranks]
    del ends
    log(f"seed shards {my_ranks} of {n_ranks}: {[round(sh.device_bytes / 2**30, 2) for sh in shards]} GiB, built in "
        f"{time.time() - t0:.1f}s")
    genome = torch.from_numpy(np.concatenate(contigs)).to(dev)
    del contigs
    use_exchange = world == 1 or dist.get_backend() == "nccl"
    if world == 1:
        xs = arcs_amd.SeedExchange.create_local(shards)
    elif use_exchange:
        uid = torch.zeros(128, dtype=torch.uint8, device=dev)
        if rank == 0:
            uid.copy_(torch.frombuffer(bytearray(arcs_amd.SeedExchange.unique_id()), dtype=torch.uint8))
        dist.broadcast(uid, 0)
        xs = [arcs_amd.SeedExchange.create(shards[0], rank, world, unique_id=bytes(uid.cpu().numpy().tobytes()))]
    else:
        xs = [None]          # gloo (ranks sharing one GPU in tests): the torch.distributed driver of arcs_amd/dist.py
    all_blocks = read_blocks(args.pairs)
    per_rank, bases = [], 0
    for r in my_ranks:
        lo, hi = blocks_of_rank(len(all_blocks), r, n_ranks)
        launches, cur, cur_pairs = [], [], 0
        for b in range(lo, hi):
            first, n = all_blocks[b]
            batch = synth.make_read_pairs(genome, n, seed=synth.SEED + 1 + b, device=dev)
            reads = arcs_amd.PackedReads.from_arrays_device(batch["ascii"], batch["offsets"], batch["lens"], device=local)
            bid = ((torch.arange(n, device=dev, dtype=torch.int64) + first) // 80).to(torch.int32)
            bases += int(batch["lens"].to(torch.int64).sum().item())
            cur.append((reads, batch["pair_ok"], bid))
            cur_pairs += n
            del batch
            if cur_pairs >= args.chunk or b == hi - 1:
                launches.append((arcs_amd.PackedReads.concat([c[0] for c in cur]), torch.cat([c[1] for c in cur]),
                                 torch.cat([c[2] for c in cur])))
                cur, cur_pairs = [], 0
        per_rank.append(launches)
    # every rank makes the same number of (collective) calls per step: ranks with fewer launches add empty ones
    n_launch = torch.tensor([max(len(l) for l in per_rank)], dtype=torch.int64, device=red_dev)
    if world > 1:
        dist.all_reduce(n_launch, op=dist.ReduceOp.MAX)
    n_launch = int(n_launch.item())
    empty = arcs_amd.PackedReads.from_arrays_device(
        torch.zeros(0, dtype=torch.uint8, device=dev), torch.zeros(1, dtype=torch.int64, device=dev),
        torch.zeros(0, dtype=torch.int32, device=dev), device=local)
    empty_ok = torch.zeros(0, dtype=torch.uint8, device=dev)
    imaps = [arcs_amd.ImapAccumulator(1 << 20, device=local) for _ in my_ranks]
    stats = [torch.zeros(8, dtype=torch.int64, device=dev) for _ in my_ranks]
    # three streams per rank: two batches in flight (submit n + 1 before complete n), the pair rule on the third
    streams = [[torch.cuda.Stream(dev) for _ in range(3)] for _ in my_ranks]

    serial = os.environ.get("ARKS_BENCH_SHARDED_SERIAL") is not None   # (profiles: one batch at a time, nothing overlaps)

    def rank_step(i, st):
        if xs[i] is not None and serial:
            with torch.cuda.stream(streams[i][0]):
                for l in range(n_launch):
                    reads, ok, bid = per_rank[i][l] if l < len(per_rank[i]) else (empty, empty_ok, None)
                    xs[i].map_pairs(reads, j, pair_ok=ok, barcode_id=bid, imap=imaps[i] if bid is not None else None, stats=st)
        elif xs[i] is not None:
            xs[i].map_pairs_pipelined(per_rank[i], j, streams[i], imap=imaps[i], stats=st, n_calls=n_launch, keep=False)
        else:
            with torch.cuda.stream(streams[i][0]):
                for l in range(n_launch):
                    reads, ok, bid = per_rank[i][l] if l < len(per_rank[i]) else (empty, empty_ok, None)
                    adist.map_pairs_seed_sharded(shards[i], reads, j, pair_ok=ok, barcode_id=bid,
                                                 imap=imaps[i] if bid is not None else None, stats=st)
        for q in streams[i]:
            q.synchronize()

    def step(with_stats=False):
        if n_local == 1:
            rank_step(0, stats[0] if with_stats else None)
            return
        errs = []

        def run(i):
            try:
                torch.cuda.set_device(local)
                rank_step(i, stats[i] if with_stats else None)
            except Exception as e:           # noqa: BLE001
                errs.append(e)
                xs[i].abort()                # the other local ranks must not wait for this one
        ts = [threading.Thread(target=run, args=(i,)) for i in range(n_local)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if errs:
            raise errs[0]

    del genome
    torch.cuda.empty_cache()         # the exchange buffers are the library's own allocations

    def mem(what):
        free, total = torch.cuda.mem_get_info(dev)
        log(f"device memory {what}: {(total - free) / 2**30:.1f} of {total / 2**30:.1f} GiB in use "
            f"(torch holds {torch.cuda.memory_reserved(dev) / 2**30:.1f})")
    mem("with the shards and the reads resident")
    for _ in range(args.warmup):
        step()
    mem("after the first pass (exchange buffers, queues)")
    step(True)
    barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        step()
    barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=red_dev)
    tot = torch.stack(stats).sum(0).to(torch.float64).to(red_dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    # the timed passes run the kernels without counters (dead reads settled per chunk: a path of its own): every
    # IndexMap count must be a multiple of the number of identical passes
    passes = args.warmup + 1 + args.steps
    timed_parity = all(bool((im.triples()[:, 2] % passes == 0).all()) for im in imaps)
    st_job = dict(zip(STAT_NAMES, [int(x) for x in tot.tolist()]))
    windows = st_job["windows"]
    if rank == 0:
        b_alg = alg_bytes_per_window(k, 279, 161)
        ms = 1e3 * elapsed / args.steps
        ex = xs[0].last_stats() if xs[0] is not None else None
        # HBM bytes of one step, every kernel of it summed (profiles/prof_sharded.sh: separate --pmc passes over this
        # very command on one GPU); with several GPUs the step's bytes divide over them like its time does
        traffic = None
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_sharded_r*.json"))):
            try:
                d = json.load(open(path))
            except (OSError, ValueError):
                continue
            if d.get("workload") == {"draft_mbp": args.draft_mbp, "pairs": args.pairs, "k": k, "shards": n_ranks, "n_gpus": 1}:
                traffic = d
        step_gb = traffic["hbm_bytes_per_step_uncalibrated"] / 1e9 if traffic else None
        # What travels between ranks, per rank and batch, from the exchange's own counts of rank 0's last batch: 8 B per
        # seed asked of another rank, 16 B per answer (arks_exchange: nothing else; tests/test_gpu_exchange.py accounts
        # for every byte over the mock transport).  Beside it the xGMI budget of SURVEY 8(e): a fully connected node
        # gives every peer pair its own link (~153 GB/s, taken as half per direction), a rank's seeds go to its 7 peers
        # evenly (hash prefix), so one link carries 1/7 of what the rank sends out and 1/7 of what it answers.
        wire = None
        if ex and n_ranks > 1:
            pairs_b = per_rank[0][-1][0].n_reads // 2 if per_rank[0] else 0
            out_b, back_b = 8 * ex["sent"], 16 * ex["sent"]
            in_b, ans_b = 8 * ex["received"], 16 * ex["received"]
            peers = n_ranks - 1
            link_dir = (out_b + ans_b) / peers                      # bytes one direction of one link carries per batch
            xgmi_dir = 153e9 / 2
            wire = {"per": "rank and batch (rank 0's last batch)", "pairs_in_batch": pairs_b,
                    "seeds": ex["seeds"], "seeds_sent": ex["sent"], "seeds_received": ex["received"],
                    "bytes_out": out_b + ans_b, "bytes_in": in_b + back_b,
                    "bytes_out_per_pair": (out_b + ans_b) / max(1, pairs_b),
                    "bytes_per_link_and_direction": link_dir,
                    "xgmi_ms_per_batch_at_76GBs_per_link_direction": 1e3 * link_dir / xgmi_dir,
                    "xgmi_pairs_per_s_per_gpu_ceiling": pairs_b / (link_dir / xgmi_dir) if link_dir else None,
                    "note": "on one GPU the local ranks read each other's buffers (no wire); the figures are what the "
                            "same batch puts on xGMI with one rank per GPU"}
        print(json.dumps({
            "metric": METRIC, "value": windows * args.steps / elapsed, "unit": "k-mers/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"synthetic {args.draft_mbp:g} Mbp draft + {args.pairs} linked-read pairs in total "
                                   f"(R1 128 / R2 151 bp), k={k} j={j}, seed table in {n_ranks} hash shards"
                                   + (f" on {world} GPU(s)" if world > 1 else f", all {n_ranks} on this one GPU "
                                      "(a local rank and host thread each)")
                                   + (" [BASELINE configs[3]]" if args.draft_mbp == 3000 and args.pairs == 500_000_000 else ""),
                       "k": k, "j": j, "shards": n_ranks, "launches_per_rank_and_step": n_launch,
                       "shard_bytes": [sh.device_bytes for sh in shards],
                       "transport": ("RCCL ncclSend/ncclRecv groups" if (world > 1 and use_exchange) else
                                     "direct: the local ranks of one process read and write each other's buffers "
                                     "(no copies); two batches in flight per rank" if world == 1 else
                                     "torch.distributed gloo through host memory (test transport)"),
                       "last_batch_of_rank0": ex, "wire": wire,
                       "parallelism": f"seed table hash-sharded x{n_ranks}, reads dealt to the ranks in blocks, seeds "
                                      "routed to their owners and back (arks_exchange)"},
            "counters": st_job, "timed_path_parity": timed_parity,
            "roofline": {"bound": "hbm", "achieved": (step_gb / world / (ms * 1e-3)) if traffic else None,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (step_gb / world / (ms * 1e-3) / HBM_PEAK_GBS) if traffic else None,
                         "traffic": (step_gb / world) if traffic else None,
                         "traffic_unit": "GB per GPU and step (rocprofv3 PMC, FETCH_SIZE x 1 + WRITE_SIZE summed over every "
                                         "kernel of the step: a lower bound, whole-line requests of the streams count half)",
                         "traffic_source": traffic["source"] if traffic else None,
                         "traffic_build_match": (traffic["kernel_build_id"] == kernel_build_id()) if traffic else None,
                         "traffic_by_kernel_GB": ({kn: round(v["fetch"] + v["write"], 2)
                                                   for kn, v in traffic["per_kernel_GB_per_step"].items()} if traffic else None),
                         "kernel": "whole step (bucket, exchange, probe, map_reads_s_kernel<REMOTE>, pair rule)",
                         "alg_achieved": windows * b_alg / (ms * 1e-3) / 1e9 / world,
                         "alg_frac": windows * b_alg / (ms * 1e-3) / 1e9 / world / HBM_PEAK_GBS,
                         "alg_bytes_per_window": b_alg}}), flush=True)
    for x in xs:
        if x is not None:
            x.close()
    assert timed_parity, "the timed passes (kernels without counters) stored other pairs than the counters pass"
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
