"""What the sharded-index configuration costs on ONE GPU: the shards of a draft are built one after the
other, the same resident batch is mapped against each (arks_map_votes_device), the votes are folded with
a maximum and checked against the whole index; so are the -v counters of the build and of the read stage (round 4:
every key is in one shard, its first holder, and the shards' counters add up).
usage: shards.py [draft Mbp] [n_shards] [pairs]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import arcs_amd
from arcs_amd import synth

mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 50.0
n_shards = int(sys.argv[2]) if len(sys.argv) > 2 else 8
pairs = int(sys.argv[3]) if len(sys.argv) > 3 else 4_000_000
k, j = 60, 0.55
contigs = synth.make_draft(int(mbp * 1e6), seed=synth.SEED)
ends = []
for c in contigs:
    cut = arcs_amd.end_cutoff(len(c))
    if cut is None:
        continue
    ends.append(c[:cut].tobytes()); ends.append(c[len(c) - cut:].tobytes())
batch = synth.make_read_pairs(contigs, pairs, seed=synth.SEED + 1, device="cuda")
reads = arcs_amd.PackedReads.from_arrays_device(batch["ascii"], batch["offsets"], batch["lens"], device=0)
w = reads.windows(k)


def timed(fn):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)


t0 = time.time()
ix = arcs_amd.ArksIndex.build(ends, k, device=0, want_stats=True)
t_whole = time.time() - t0
whole_build = ix.build_stats
whole_st = torch.zeros(8, dtype=torch.int64, device="cuda")
arcs_amd.map_reads_packed(ix, reads, j, stats=whole_st)
whole_st = whole_st.cpu().tolist()
whole = arcs_amd.map_votes_packed(ix, reads).clone()
ms_whole = timed(lambda: arcs_amd.map_votes_packed(ix, reads))
ms_plain = timed(lambda: arcs_amd.map_reads_packed(ix, reads, j))
plain = arcs_amd.map_reads_packed(ix, reads, j).clone()
print(f"whole index: {len(ix)} keys, {ix.device_bytes / 2**20:.0f} MiB, built in {t_whole:.2f} s; "
      f"map {ms_plain:.2f} ms, votes {ms_whole:.2f} ms ({w / ms_whole / 1e6:.1f} G k-mers/s)", flush=True)
ix.close()
votes = None
build_sum, parts, keys_sum = {}, [], 0
for s in range(n_shards):
    t0 = time.time()
    sh = arcs_amd.ArksIndex.build_shard(ends, k, s, n_shards, device=0, want_stats=True)
    tb = time.time() - t0
    for f, x in sh.build_stats.items():
        build_sum[f] = build_sum.get(f, 0) + x
    keys_sum += len(sh)
    t = torch.zeros(8, dtype=torch.int64, device="cuda")
    arcs_amd.map_reads_packed(sh, reads, j, stats=t)
    parts.append(t.cpu().tolist())
    v = arcs_amd.map_votes_packed(sh, reads).clone()
    ms = timed(lambda: arcs_amd.map_votes_packed(sh, reads))
    votes = v if votes is None else torch.maximum(votes, v)
    print(f"shard {s}/{n_shards}: {len(sh)} keys, {sh.device_bytes / 2**20:.0f} MiB, built in {tb:.2f} s; "
          f"votes {ms:.2f} ms ({w / ms / 1e6:.1f} G k-mers/s)", flush=True)
    sh.close()
got = arcs_amd.resolve_votes(votes, reads, k, j)
print("max of shard votes == whole index:", bool(torch.equal(votes, whole)),
      " resolved == plain map:", bool(torch.equal(got, plain)), flush=True)
folded = torch.zeros(8, dtype=torch.int64, device="cuda")
arcs_amd.count_votes(votes, reads, k, j, folded)
g = folded.cpu().tolist()
g[2:5] = [sum(p[i] for p in parts) for i in (2, 3, 4)]
g[0], g[1], g[7] = parts[0][0], parts[0][1], parts[0][7]
print("build counters, sum over the shards == whole index:", build_sum == whole_build, build_sum)
print("keys held by the shards in all:", keys_sum, "== keys of the whole index:", keys_sum == whole_build["recorded"])
print("read-stage counters folded over the shards == whole index:", g == whole_st, g)
