"""Randomised differential test: GPU path vs CPU oracle over many (k, draft, read shape) combinations."""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import arcs_amd
from arcs_amd import synth
from oracle import pyoracle as oracle
oracle.build_oracle()
COMP = str.maketrans("ACGTacgtNn", "TGCAtgcaNn")
def rc(s): return s[::-1].translate(COMP)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120
t_end = time.time() + budget
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n_cases = n_reads_total = 0
while time.time() < t_end:
    rng = np.random.Generator(np.random.PCG64(seed)); seed += 1
    if os.environ.get("FUZZ_TRACE"):                      # the case under way, for a run that dies in a kernel
        open(os.environ["FUZZ_TRACE"], "w").write(str(seed - 1))
    k = int(rng.choice([20, 21, 22, 23, 24, 25, 27, 30, 31, 32, 33, 40, 45, 59, 60, 61, 63, 64, 65, 72, 80, 95, 96]))
    def rnd(n): return "".join("ACGT"[i] for i in rng.integers(0, 4, size=n))
    ends = []
    base = rnd(int(rng.integers(2000, 20000)))
    for e in range(int(rng.integers(2, 12))):
        L = int(rng.choice([64, 96, 500, 1000, 3000, 320, 640, 2048]))
        mode = int(rng.integers(0, 6))
        if mode == 0 and len(base) > L:      # shares a segment with `base` (value 0 keys, second diagonals)
            p = int(rng.integers(0, len(base) - L)); s = base[p:p + L]
        elif mode == 1:                      # low complexity
            u = rnd(int(rng.integers(1, 40))); s = (u * (L // len(u) + 1))[:L]
        elif mode == 2:                      # palindromic stretch
            h = rnd(L // 2); s = h + rc(h)
        else:
            s = rnd(L)
        s = list(s)
        for q in rng.integers(0, L, size=int(rng.integers(0, 4))): s[q] = "N"
        if rng.random() < 0.2 and L > 200: s[100:100 + int(rng.integers(2, 150))] = "N" * len(s[100:100 + int(rng.integers(2, 150))])
        ends.append("".join(s))
    ends.append(base)
    ox = oracle.OracleIndex(k).build(ends)
    # both layouts of the locality index take turns (arks_build_options through api.BUILD_DEFAULTS: the library reads no
    # environment any more, and this process no longer calls setenv with the HIP runtime's threads alive)
    arcs_amd.api.BUILD_DEFAULTS["index_kind"] = "seeds" if seed % 3 else "minimizer"
    # every other case: the medium kernel on a few waves only, so that its (short) queue is taken several reads per
    # grab -- tiles of several gathered reads
    arcs_amd.api.set_medium_blocks(1 + seed % 5 if seed % 2 else 0)
    # two cases in five: m-mers heavy beyond 8 occurrences instead of 2 (seeds with 3-8 entries, whose windows take the
    # walk over the entries)
    arcs_amd.api.BUILD_DEFAULTS["heavy_over"] = 8 if seed % 5 < 2 else 0
    ix = arcs_amd.ArksIndex.build(ends, k, device=0)
    assert {f: ix.build_stats[f] for f in ox.stats.as_dict()} == ox.stats.as_dict(), (seed, k, "build stats")
    genome = "".join(ends)
    reads = []
    for i in range(int(rng.integers(200, 1500))):
        L = int(rng.choice([k - 1, k, k + 1, 31, 32, 33, 64, 100, 128, 150, 151, 250, 300, 511, 512, 513, 700, 1]))
        L = max(0, min(L, len(genome) - 1))
        p = int(rng.integers(0, len(genome) - L))
        r = list(genome[p:p + L])
        if rng.random() < 0.15 and L > 2 * k:                      # chimera: second half from elsewhere
            p2 = int(rng.integers(0, len(genome) - L)); r[L // 2:] = genome[p2 + L // 2:p2 + L]
        for q in rng.integers(0, max(L, 1), size=int(rng.integers(0, 4)) if L else 0):
            r[q] = "ACGTNacgtn"[int(rng.integers(10))]
        r = "".join(r)
        reads.append(rc(r) if i % 2 else r)
    for j in (0.55, 0.0, float(rng.random())):
        st = oracle.MapStats()
        want = [ox.best_contig(r, j, st) for r in reads]
        got, gst = ix.map_reads(reads, j, want_stats=True)
        bad = [i for i, (a, b) in enumerate(zip(got.tolist(), want)) if a != b]
        assert not bad, (seed - 1, k, j, bad[:5], [len(reads[i]) for i in bad[:5]])
        assert gst == st.as_dict(), (seed - 1, k, j, gst, st.as_dict())
        # the kernels WITHOUT counters are instantiations of their own (shortcuts for reads without seed entries)
        assert ix.map_reads(reads, j).tolist() == want, (seed - 1, k, j, "no counters")
    if os.environ.get("FUZZ_SHARDS"):      # the same reads through 2..5 index shards: votes, maximum, j_index test
        n_sh = int(rng.integers(2, 6))
        packed = arcs_amd.PackedReads.from_ascii(reads, device=0)
        votes = None
        share = {}
        parts = []
        for sidx in range(n_sh):
            sh = arcs_amd.ArksIndex.build_shard(ends, k, sidx, n_sh, device=0, want_stats=True)
            for f, v in sh.build_stats.items():
                share[f] = share.get(f, 0) + v
            v = arcs_amd.map_votes_packed(sh, packed).clone()
            votes = v if votes is None else arcs_amd.max_votes(votes, v)
            t = torch.zeros(8, dtype=torch.int64, device="cuda")     # the read stage's counters of this shard
            arcs_amd.map_reads_packed(sh, packed, 0.55, stats=t)
            parts.append(t.cpu().tolist())
            torch.cuda.synchronize(); sh.close()
        # the shards' shares of the build counters add up to the serial loop's over all the ends
        assert {f: share[f] for f in ox.stats.as_dict()} == ox.stats.as_dict(), (seed - 1, k, n_sh, "build counters")
        assert torch.equal(votes, arcs_amd.map_votes_packed(ix, packed)), (seed - 1, k, n_sh, "votes")
        for j in (0.55, 0.0):
            got = arcs_amd.resolve_votes(votes, packed, k, j).cpu().tolist()
            st2 = oracle.MapStats()
            want = [ox.best_contig(r, j, st2) for r in reads]
            bad = [i for i, (a, b) in enumerate(zip(got, want)) if a != b]
            assert not bad, (seed - 1, k, j, n_sh, "sharded", bad[:5])
            # every key is in one shard (its first holder): found, recorded, duplicates add up; the other window
            # counters are the same in every shard; the j_index test is counted on the folded votes
            folded = torch.zeros(8, dtype=torch.int64, device="cuda")
            arcs_amd.count_votes(votes, packed, k, j, folded)
            g = folded.cpu().tolist()
            g[2:5] = [sum(p[i] for p in parts) for i in (2, 3, 4)]
            g[0], g[1], g[7] = parts[0][0], parts[0][1], parts[0][7]
            assert all(p[0] == g[0] and p[1] == g[1] and p[7] == g[7] for p in parts), (seed - 1, k, n_sh, "window counters")
            assert dict(zip(st2.as_dict(), g)) == st2.as_dict(), (seed - 1, k, j, n_sh, "read-stage counters", g, st2.as_dict())
    if os.environ.get("FUZZ_SEED_SHARDS"):  # the seed table in 2..4 shards, every seed answered by its owner
        n_sh = int(rng.integers(2, 5))
        packed = arcs_amd.PackedReads.from_ascii(reads, device=0)
        shards = [arcs_amd.ArksIndex.build_seed_shard(ends, k, r, n_sh, device=0) for r in range(n_sh)]
        counts = arcs_amd.api.seed_counts(shards[0], packed)
        seed_off = torch.zeros(packed.n_reads + 1, dtype=torch.int64, device="cuda")
        seed_off[1:] = torch.cumsum(counts.to(torch.int64), 0)
        mmer, owner = arcs_amd.api.seeds_fill(shards[0], packed, seed_off)
        answers = torch.zeros(2 * mmer.numel(), dtype=torch.int64, device="cuda")
        for r, sh in enumerate(shards):
            sel = (owner == r).nonzero().flatten()
            answers.view(-1, 2)[sel] = arcs_amd.api.seeds_probe(sh, mmer[sel].contiguous()).view(-1, 2)
        for j in (0.55, 0.0):
            stats = torch.zeros(8, dtype=torch.int64, device="cuda")
            got = arcs_amd.api.map_reads_seeded(shards[int(rng.integers(n_sh))], packed, j, seed_off, answers, stats=stats).cpu().tolist()
            st = oracle.MapStats()
            want = [ox.best_contig(r, j, st) for r in reads]
            bad = [i for i, (a, b) in enumerate(zip(got, want)) if a != b]
            assert not bad, (seed - 1, k, j, n_sh, "seed shards", bad[:5])
            assert dict(zip(("total_valid", "bad", "found", "recorded", "dups", "reads_pass", "reads_fail", "windows"), stats.cpu().tolist())) == st.as_dict(), (seed - 1, k, j, n_sh)
        if shards[0].kind == 2:
            # the product path over the same shards: arks_exchange, the ranks of a local group driven by this one thread
            # (submit per rank, arks_exchange_complete_group), the reads dealt unevenly, with and without counters
            xs = arcs_amd.SeedExchange.create_local(shards)
            cuts = sorted(int(x) for x in rng.integers(0, len(reads) + 1, size=n_sh - 1))
            cuts = [0] + cuts + [len(reads)]
            parts = [arcs_amd.PackedReads.from_ascii(reads[cuts[r]:cuts[r + 1]], device=0) for r in range(n_sh)]
            jx = float(rng.choice([0.55, 0.0, 0.3]))
            want = [ox.best_contig(r, jx) for r in reads]
            for with_stats in (False, True):
                sts = [torch.zeros(8, dtype=torch.int64, device="cuda") for _ in range(n_sh)]
                outs = [xs[r].submit(parts[r], jx, stats=sts[r] if with_stats else None) for r in range(n_sh)]
                arcs_amd.SeedExchange.complete_group(xs)
                torch.cuda.synchronize()
                got = sum([o.cpu().tolist()[:parts[r].n_reads] for r, o in enumerate(outs)], [])
                bad = [i for i, (a, b) in enumerate(zip(got, want)) if a != b]
                assert not bad, (seed - 1, k, jx, n_sh, "exchange", with_stats, bad[:5])
            for x in xs:
                x.close()
        for sh in shards:
            sh.close()
    ix.close()
    n_cases += 1; n_reads_total += len(reads)
# Orderly end: everything of the library is closed above, the device is idle, the verdict is flushed -- and then the
# process leaves WITHOUT the interpreter's and the runtimes' exit-time teardown (torch's and the HIP runtime's static
# destructors with their helper threads still alive).  Round 5's driver run lost one of twelve such children to a
# SIGSEGV that left no HIP message and no python traceback; 372 + 400 process-runs since did not reproduce it (DESIGN
# section 9), and teardown at exit is the one phase of such a child that is not the library's.  FUZZ_TRACE says which
# phase a child was in, should it happen again: "<case>" while a case runs, "done" after the last one, "exit" here.
torch.cuda.synchronize()
if os.environ.get("FUZZ_TRACE"):
    open(os.environ["FUZZ_TRACE"], "w").write("done after %d" % (seed - 1))
print("fuzz ok: %d cases, %d reads, last seed %d" % (n_cases, n_reads_total, seed - 1), flush=True)
if os.environ.get("FUZZ_TRACE"):
    open(os.environ["FUZZ_TRACE"], "w").write("exit after %d" % (seed - 1))
sys.stdout.flush(); sys.stderr.flush()
if not os.environ.get("FUZZ_FULL_TEARDOWN"):
    os._exit(0)
