"""BASELINE config 3 on one MI355X: 3 Gbp draft + 500 M linked-read pairs, k=60, j=0.55.
The reads are generated on the device in chunks (they do not fit as ASCII), each chunk is packed,
gated, mapped and accumulated into ONE IndexMap; reports the index build, the mapping time over
all chunks (HIP events around the map stage only) and a digest of the final triples."""
import hashlib, json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import arcs_amd
from arcs_amd import synth
MBP = float(os.environ.get("C3_MBP", 3000)); PAIRS = int(os.environ.get("C3_PAIRS", 500_000_000)); CH = int(os.environ.get("C3_CHUNK", 20_000_000))
t0 = time.time()
contigs = synth.make_draft(int(MBP * 1e6), seed=synth.SEED)
ends = []
for c in contigs:
    cut = arcs_amd.end_cutoff(len(c))
    if cut is None: continue
    ends.append(c[:cut].tobytes()); ends.append(c[len(c) - cut:].tobytes())
t1 = time.time()
ix = arcs_amd.ArksIndex.build(ends, 60, device=0, want_stats=True)
torch.cuda.synchronize(); t_build = time.time() - t1
del ends
print("draft %.0f Mbp, %d contigs; index %d keys, %.2f GiB, built in %.1f s" % (MBP, len(contigs), len(ix), ix.device_bytes / 2**30, t_build), flush=True)
imap = arcs_amd.ImapAccumulator(1 << 27, device=0)
stored = torch.zeros(1, dtype=torch.int64, device="cuda")
stats = torch.zeros(8, dtype=torch.int64, device="cuda")
map_ms = 0.0; windows = 0; done = 0; t2 = time.time()
while done < PAIRS:
    n = min(CH, PAIRS - done)
    batch = synth.make_read_pairs(contigs, n, seed=synth.SEED + 1 + done // CH, device="cuda")
    reads = arcs_amd.PackedReads.from_arrays_device(batch["ascii"], batch["offsets"], batch["lens"], device=0)
    bid = batch["barcode_id"] + (done // 80)          # barcodes continue across chunks
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    arcs_amd.map_pairs_packed(ix, reads, 0.55, pair_ok=batch["pair_ok"], barcode_id=bid.to(torch.int32), imap=imap, stats=(None if os.environ.get('C3_NOSTATS') else stats), stored=stored)
    b.record(); torch.cuda.synchronize()
    map_ms += a.elapsed_time(b); windows += reads.windows(60); done += n
    del batch, reads
    print("  %d pairs done, map stage %.1f ms so far, wall %.0f s" % (done, map_ms, time.time() - t2), flush=True)
tr = imap.triples()
digest = hashlib.sha256(np.ascontiguousarray(tr).tobytes()).hexdigest()[:16]
res = dict(config="BASELINE configs[2]: 3 Gbp draft + 500 M pairs, k=60 j=0.55, 1x MI355X", draft_mbp=MBP, pairs=done, windows=windows,
           index_keys=len(ix), index_gib=ix.device_bytes / 2**30, index_build_s=t_build, map_stage_ms=map_ms,
           kmers_per_s=windows / map_ms * 1e3, stored_pairs=int(stored.item()), triples=int(tr.shape[0]), triples_sha256_16=digest,
           counters=dict(zip(("total_valid", "bad", "found", "recorded", "dups", "reads_pass", "reads_fail", "windows"), stats.cpu().tolist())),
           wall_s_including_generation=time.time() - t0)
print(json.dumps(res))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/c3_run.json", "w"), indent=1)
