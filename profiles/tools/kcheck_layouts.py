"""map a small batch at one k with one index layout and compare with the oracle (run per k in a subprocess)"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
k = int(sys.argv[1]); import arcs_amd.api; arcs_amd.api.BUILD_DEFAULTS["index_kind"] = sys.argv[2]
import arcs_amd
from arcs_amd import synth
from oracle import pyoracle as O
j = 0.5
contigs = synth.make_draft(2_000_000, seed=91)
cs = synth.contigs_to_strings(contigs)
ix = arcs_amd.ArksIndex.build(arcs_amd.contig_ends(cs), k, device=0)
print("k", k, "kind", ix.kind, "keys", len(ix), "bytes", ix.device_bytes, flush=True)
ox = O.OracleIndex(k).build(O.contig_ends(cs))
assert {f: ix.build_stats[f] for f in ox.stats.as_dict()} == ox.stats.as_dict()
batch = synth.make_read_pairs(contigs, int(sys.argv[3]) if len(sys.argv) > 3 else 5000, seed=92, device="cuda")
reads = arcs_amd.PackedReads.from_arrays_device(batch["ascii"], batch["offsets"], batch["lens"], device=0)
stats = torch.zeros(8, dtype=torch.int64, device="cuda")
c, p = arcs_amd.map_pairs_packed(ix, reads, j, pair_ok=batch["pair_ok"], stats=stats)
torch.cuda.synchronize()
a = np.concatenate([batch["ascii"].cpu().numpy(), np.zeros(1, np.uint8)])
wc, wp, wst = ox.map_pairs(a, batch["offsets"].cpu().numpy().astype(np.uint64)[:-1], batch["lens"].cpu().numpy().astype(np.uint32), j, pair_ok=batch["pair_ok"].cpu().numpy(), threads=16)
got = dict(zip(("total_valid", "bad", "found", "recorded", "dups", "reads_pass", "reads_fail", "windows"), stats.cpu().tolist()))
print("conreci equal", bool((c.cpu().numpy() == wc).all()), "pairs equal", bool((p.cpu().numpy() == wp).all()), "stats equal", all(got[f] == wst[f] for f in got), flush=True)
