import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import arcs_amd
from arcs_amd import synth
mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 1000.0
t0 = time.time()
contigs = synth.make_draft(int(mbp * 1e6), seed=synth.SEED)
print("draft", len(contigs), "contigs", time.time() - t0, flush=True)
ends = []
for c in contigs:
    cut = arcs_amd.end_cutoff(len(c))
    if cut is None: continue
    ends.append(c[:cut].tobytes()); ends.append(c[len(c) - cut:].tobytes())
print("ends", len(ends), sum(map(len, ends)) / 1e6, "Mbp", time.time() - t0, flush=True)
t1 = time.time()
ix = arcs_amd.ArksIndex.build(ends, 60, device=0, want_stats=True)
torch.cuda.synchronize()
print("index built in", time.time() - t1, "s:", len(ix), "keys,", ix.device_bytes / 2**30, "GiB", ix.build_stats, flush=True)
print("torch mem", torch.cuda.memory_allocated() / 2**30, "free/total", [x / 2**30 for x in torch.cuda.mem_get_info()], flush=True)
del ends
batch = synth.make_read_pairs(contigs, 4_000_000, seed=synth.SEED + 1, device="cuda")
reads = arcs_amd.PackedReads.from_arrays_device(batch["ascii"], batch["offsets"], batch["lens"], device=0)
step = arcs_amd.PairStep(ix, reads, 0.55, pair_ok=batch["pair_ok"], barcode_id=batch["barcode_id"])
step.run(); torch.cuda.synchronize()
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record(); step.run(); b.record(); torch.cuda.synchronize()
w = reads.windows(60)
print("map: %.2f ms  %.2f G k-mers/s  pass reads %d" % (a.elapsed_time(b), w / a.elapsed_time(b) / 1e6, int((step.conreci != 0).sum())), flush=True)
