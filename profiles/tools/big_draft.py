"""A draft whose contig-end text exceeds what ONE index can address (2^32 text positions; the exact hash
table it would fall back to needs 64 B per k-mer of scratch and does not fit 288 GB either), mapped on
one MI355X through index shards.  No whole index exists to compare with, so the check is the
size-independent property: the results with 2 shards and with 3 shards are exact, hence identical.
usage: big_draft.py [draft Mbp = 4600] [pairs = 4000000]"""
import ctypes as C, hashlib, json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import arcs_amd
from arcs_amd import synth
from arcs_amd._lib import check, lib

mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 4600.0
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
k, j, END = 60, 0.55, 120000          # -e 120000: the ends cover every contig completely
t0 = time.time()
contigs = synth.make_draft(int(mbp * 1e6), seed=synth.SEED)
parts, lens = [], []
for c in contigs:
    cut = arcs_amd.end_cutoff(len(c), 500, END)
    if cut is None:
        continue
    parts.append(c[:cut]); parts.append(c[len(c) - cut:])
    lens += [cut, cut]
lens = np.array(lens, dtype=np.uint32)
offs = np.zeros(len(lens) + 1, dtype=np.uint64)
np.cumsum(lens, out=offs[1:])
data = np.concatenate(parts + [np.zeros(1, np.uint8)])
del parts
print(f"draft {mbp:.0f} Mbp, {len(contigs)} contigs, {len(lens)} ends, {int(offs[-1]) / 1e9:.2f} G end positions "
      f"(2^32 = 4.29 G) in {time.time() - t0:.0f} s", flush=True)
batch = synth.make_read_pairs(contigs, pairs, seed=synth.SEED + 1, device="cuda")
reads = arcs_amd.PackedReads.from_arrays_device(batch["ascii"], batch["offsets"], batch["lens"], device=0)
w = reads.windows(k)
ev = arcs_amd.pair_gate(reads, batch["pair_ok"])
out = {"draft_mbp": mbp, "end_positions": int(offs[-1]), "pairs": pairs, "windows": w, "k": k, "runs": []}
digests = []
for n_shards in (2, 3):
    shards, t_build = [], []
    for s in range(n_shards):
        t1 = time.time()
        h = C.c_void_p()
        check(lib().arks_index_build_shard(C.byref(h), k, data.ctypes.data, offs.ctypes.data, lens.ctypes.data,
                                           len(lens), s, n_shards, 0), "arks_index_build_shard")
        t_build.append(time.time() - t1)
        shards.append(arcs_amd.ArksIndex(h, k, 0, None))
        print(f"  {n_shards} shards: shard {s}: {len(shards[-1])} keys, kind {shards[-1].kind}, "
              f"{shards[-1].device_bytes / 2**30:.2f} GiB, built in {t_build[-1]:.1f} s", flush=True)
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    votes = None
    for sh in shards:
        v = arcs_amd.map_votes_packed(sh, reads, eval_mask=ev)
        votes = v.clone() if votes is None else arcs_amd.max_votes(votes, v)
    conreci = arcs_amd.resolve_votes(votes, reads, k, j)
    pair = arcs_amd.pairs_rule(conreci, reads, batch["pair_ok"])
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b)
    d = hashlib.sha256(conreci.cpu().numpy().tobytes() + pair.cpu().numpy().tobytes()).hexdigest()[:16]
    digests.append(d)
    stored = int((pair != 0).sum().item())
    print(f"  {n_shards} shards: map stage {ms:.1f} ms ({w / ms / 1e6:.1f} G k-mers/s), reads mapped "
          f"{int((conreci != 0).sum().item())}, pairs stored {stored}, digest {d}", flush=True)
    out["runs"].append({"n_shards": n_shards, "keys": [len(x) for x in shards], "kinds": [x.kind for x in shards],
                        "index_gib": [x.device_bytes / 2**30 for x in shards], "build_s": t_build, "map_ms": ms,
                        "kmers_per_s": w / ms * 1e3, "pairs_stored": stored, "digest": d})
    for sh in shards:
        sh.close()
out["identical"] = len(set(digests)) == 1
print("2 shards == 3 shards:", out["identical"], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/big_draft.json", "w"), indent=1)
