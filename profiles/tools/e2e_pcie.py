"""PCIe-inclusive rate: packed read batches start in pinned HOST memory; per batch H2D (codes, nmask,
offsets, lengths, class, pair_ok, barcode ids) -> gate/map/pairs -> D2H (pair results); two streams
double-buffer the batches.  (bench.py's `value` is the HBM-resident rate; this is the DESIGN.md note.)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import arcs_amd
from arcs_amd import synth
k, j = 60, 0.55
NB, PAIRS = 8, 2_000_000
contigs = synth.make_draft(50_000_000, seed=synth.SEED)
ix = arcs_amd.ArksIndex.build(arcs_amd.contig_ends(synth.contigs_to_strings(contigs)), k, device=0, want_stats=False)
host = []
for b in range(2):   # two distinct host batches, reused
    batch = synth.make_read_pairs(contigs, PAIRS, seed=100 + b, device="cuda")
    r = arcs_amd.PackedReads.from_arrays_device(batch["ascii"], batch["offsets"], batch["lens"], device=0)
    host.append({n: t.cpu().pin_memory() for n, t in
                 dict(codes=r.codes, nmask=r.nmask, woff=r.word_off, lens=r.lens, cls=r.read_class,
                      ok=batch["pair_ok"], bid=batch["barcode_id"]).items()})
    windows = r.windows(k)
    del batch, r
torch.cuda.synchronize()
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
dev = [{n: torch.empty_like(t, device="cuda") for n, t in host[0].items()} for _ in range(2)]
outs = [torch.empty(PAIRS, dtype=torch.int32).pin_memory() for _ in range(2)]
steps = []
for s in range(2):
    d = dev[s]
    reads = arcs_amd.PackedReads(d["codes"], d["nmask"], d["woff"], d["lens"], d["cls"], 0)
    steps.append(arcs_amd.PairStep(ix, reads, j, pair_ok=d["ok"], barcode_id=d["bid"]))
bytes_in = sum(t.numel() * t.element_size() for t in host[0].values())
def run(nb):
    for b in range(nb):
        s = b & 1
        with torch.cuda.stream(streams[s]):
            for n, t in host[s].items():
                dev[s][n].copy_(t, non_blocking=True)
            steps[s].run()
            outs[s].copy_(steps[s].pair[:PAIRS], non_blocking=True)
    torch.cuda.synchronize()
run(2)
t0 = time.perf_counter(); run(NB); dt = time.perf_counter() - t0
print(f"PCIe-inclusive: {NB * windows / dt / 1e9:.2f} G k-mers/s  ({NB} batches of {PAIRS} pairs, {bytes_in / 1e6:.0f} MB in per batch, "
      f"{NB * bytes_in / dt / 1e9:.1f} GB/s H2D, {dt / NB * 1e3:.1f} ms per batch)")
