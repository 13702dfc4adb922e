"""What the sharded-index configuration costs on ONE GPU: the shards of a draft are built one after the
other, the same resident batch is mapped against each (arks_map_votes_device), the votes are folded with
a maximum and checked against the whole index.  usage: shards.py [draft Mbp] [n_shards] [pairs]"""
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
ix = arcs_amd.ArksIndex.build(ends, k, device=0, want_stats=False)
t_whole = time.time() - t0
whole = arcs_amd.map_votes_packed(ix, reads).clone()
ms_whole = timed(lambda: arcs_amd.map_votes_packed(ix, reads))
ms_plain = timed(lambda: arcs_amd.map_reads_packed(ix, reads, j))
plain = arcs_amd.map_reads_packed(ix, reads, j).clone()
print(f"whole index: {len(ix)} keys, {ix.device_bytes / 2**20:.0f} MiB, built in {t_whole:.2f} s; "
      f"map {ms_plain:.2f} ms, votes {ms_whole:.2f} ms ({w / ms_whole / 1e6:.1f} G k-mers/s)", flush=True)
ix.close()
votes = None
for s in range(n_shards):
    t0 = time.time()
    sh = arcs_amd.ArksIndex.build_shard(ends, k, s, n_shards, device=0)
    tb = time.time() - t0
    v = arcs_amd.map_votes_packed(sh, reads).clone()
    ms = timed(lambda: arcs_amd.map_votes_packed(sh, reads))
    votes = v if votes is None else torch.maximum(votes, v)
    print(f"shard {s}/{n_shards}: {len(sh)} keys, {sh.device_bytes / 2**20:.0f} MiB, built in {tb:.2f} s; "
          f"votes {ms:.2f} ms ({w / ms / 1e6:.1f} G k-mers/s)", flush=True)
    sh.close()
got = arcs_amd.resolve_votes(votes, reads, k, j)
print("max of shard votes == whole index:", bool(torch.equal(votes, whole)),
      " resolved == plain map:", bool(torch.equal(got, plain)), flush=True)
