"""Several processes on one GPU, each building an index through the build's address-range arena (VmArena: HIP's
virtual-memory API) and freeing it, over and over, mapping the same reads each time: every digest must be the same.
usage: arena_stress.py <processes> <rounds>"""
import os, subprocess, sys
sys.path.insert(0, os.getcwd())
if len(sys.argv) > 3:          # a child
    import torch
    import arcs_amd
    from arcs_amd import synth
    rounds = int(sys.argv[2])
    contigs = synth.make_draft(20_000_000, seed=77)
    ends = arcs_amd.contig_ends(synth.contigs_to_strings(contigs))
    batch = synth.make_read_pairs(contigs, 200_000, seed=78, device="cuda")
    reads = arcs_amd.PackedReads.from_arrays_device(batch["ascii"], batch["offsets"], batch["lens"], device=0)
    digs = set()
    for r in range(rounds):
        ix = arcs_amd.ArksIndex.build(ends, 60, device=0, want_stats=(r % 2 == 0))
        assert ix.kind == 2
        c, p = arcs_amd.map_pairs_packed(ix, reads, 0.55, pair_ok=batch["pair_ok"])
        w = torch.arange(c.numel(), device="cuda", dtype=torch.int64) % 1000003 + 1
        digs.add((int((c.to(torch.int64) * w).sum().item()), int((p != 0).sum().item()), len(ix)))
        ix.close()
    print("digests", sorted(digs), flush=True)
    sys.exit(0 if len(digs) == 1 else 1)
n, rounds = int(sys.argv[1]), int(sys.argv[2])
ps = [subprocess.Popen([sys.executable, __file__, str(n), str(rounds), "child"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for _ in range(n)]
outs = [p.communicate() for p in ps]
lines = {o.strip().splitlines()[-1] if o.strip() else "?" for o, _ in outs}
print("return codes", [p.returncode for p in ps]); print(lines)
for (o, e), p in zip(outs, ps):
    if p.returncode:
        print(e[-1500:])
sys.exit(0 if all(p.returncode == 0 for p in ps) and len(lines) == 1 else 1)
