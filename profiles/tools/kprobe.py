import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
import torch, numpy as np, arcs_amd
from arcs_amd import synth, _lib
k=int(sys.argv[1]); mbp=float(sys.argv[2]) if len(sys.argv)>2 else 50
contigs = synth.make_draft(int(mbp*1e6), seed=synth.SEED)
cs = synth.contigs_to_strings(contigs)
ix = arcs_amd.ArksIndex.build(arcs_amd.contig_ends(cs), k, device=0, want_stats=False)
RL=int(os.environ.get("RL","0"))
batch = synth.make_read_pairs(contigs, 4_000_000, seed=synth.SEED + 1, device="cuda", **(dict(r1_len=RL, r2_len=RL, frag=2*RL) if RL else {}))
reads = arcs_amd.PackedReads.from_arrays_device(batch["ascii"], batch["offsets"], batch["lens"], device=0)
step = arcs_amd.PairStep(ix, reads, 0.55, pair_ok=batch["pair_ok"], barcode_id=batch["barcode_id"])
step.run(); torch.cuda.synchronize()
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record(); step.run(); b.record(); torch.cuda.synchronize()
q = (C.c_uint * 4)()
_lib.lib().arks_debug_queue_counts(ix.handle, q)
print(k, os.environ.get("ARKS_MINIMIZER_LEN","-"), "kind", ix.kind, "%.2f ms" % a.elapsed_time(b), "%.1f G/s" % (reads.windows(k) / a.elapsed_time(b) / 1e6), "slow", q[0], "medium", q[2], "of", reads.n_reads, "index GiB %.2f" % (ix.device_bytes / 2**30), flush=True)
