import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import arcs_amd
from arcs_amd import synth, api
k, j = 60, 0.55
contigs = synth.make_draft(100_000_000, seed=synth.SEED)
ends = []
for c in contigs:
    cut = arcs_amd.end_cutoff(len(c))
    if cut is not None:
        ends += [c[:cut].tobytes(), c[len(c) - cut:].tobytes()]
ix = arcs_amd.ArksIndex.build_seed_shard(ends, k, 0, 1, device=0)
genome = torch.from_numpy(np.concatenate(contigs)).cuda()
batch = synth.make_read_pairs(genome, 20_000_000, seed=5, device="cuda")
reads = arcs_amd.PackedReads.from_arrays_device(batch["ascii"], batch["offsets"], batch["lens"], device=0)
ev = api.pair_gate(reads, batch["pair_ok"])
def t(f, n=3):
    torch.cuda.synchronize(); s = time.time()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.time() - s) / n * 1e3, r
ms, counts = t(lambda: api.seed_counts(ix, reads, ev)); print("seed_counts %.2f ms" % ms)
def pre():
    so = torch.zeros(reads.n_reads + 1, dtype=torch.int64, device="cuda"); so[1:] = torch.cumsum(counts.to(torch.int64), 0); return so
ms, seed_off = t(pre); print("cumsum %.2f ms" % ms)
ms, (mmer, owner) = t(lambda: api.seeds_fill(ix, reads, seed_off, ev)); print("seeds_fill %.2f ms (%d seeds)" % (ms, mmer.numel()))
own8 = (mmer % 8).to(torch.int32)
ms, order = t(lambda: torch.argsort(own8.to(torch.int64), stable=True)); print("argsort(int64, stable) %.2f ms" % ms)
ms, _ = t(lambda: torch.sort(own8.to(torch.uint8), stable=True)); print("sort(uint8, stable) %.2f ms" % ms)
ms, send = t(lambda: mmer[order].contiguous()); print("gather %.2f ms" % ms)
ms, _ = t(lambda: torch.bincount(own8.to(torch.int64), minlength=8)); print("bincount %.2f ms" % ms)
ms, ans = t(lambda: api.seeds_probe(ix, mmer)); print("seeds_probe %.2f ms" % ms)
def unperm():
    out = torch.empty_like(ans); out.view(-1, 2)[order] = ans.view(-1, 2); return out
ms, _ = t(unperm); print("unpermute %.2f ms" % ms)
ms, _ = t(lambda: api.map_reads_seeded(ix, reads, j, seed_off, ans, eval_mask=ev)); print("map_reads_seeded %.2f ms" % ms)
ix2 = arcs_amd.ArksIndex.build(ends, k, device=0)
ms, _ = t(lambda: arcs_amd.map_reads_packed(ix2, reads, j, eval_mask=ev)); print("map_reads (fused) %.2f ms" % ms)
