"""Multi-GPU driver logic (one process per GPU, torch.distributed; backend "nccl" = RCCL on ROCm,
"gloo" in the CPU tests).  The path shards over reads: every rank holds a replica of the contig
k-mer index and maps its own slice of the read pairs; the only exchange is the final merge of the
per-rank IndexMap triples (a sum), which mirrors `imap[barcode][end]++` being commutative
(Arcs/Arcs.cpp:1282-1285).  No data-path collective.

The sharded-index configuration (BASELINE configs[3]; a draft whose index should not live on one GPU)
is the second half of this file: rank r holds shard r of the index (arks_index_build_shard), every
rank maps the SAME read batch against its shard, and the one data-path collective is an
all-reduce(MAX) of the 8-byte per-read votes -- see include/arks_hip.h, arks_map_votes_device."""
import numpy as np


def shard_pairs(n_pairs, rank, world):
    """contiguous slice [lo, hi) of the pairs that rank maps (mates stay together)"""
    base, rem = divmod(n_pairs, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def merge_triples(local, group=None):
    """all ranks contribute uint32[n, 3] (barcode id, conreci, count); every rank gets the merged,
    sorted sum.  Sizes differ per rank, so sizes are gathered first and the payload is padded."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    loc = torch.from_numpy(np.ascontiguousarray(local, dtype=np.uint32).astype(np.int64)).to(dev)
    n = torch.tensor([loc.shape[0]], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    m = int(max(int(x.item()) for x in sizes))
    pad = torch.zeros((m, 3), dtype=torch.int64, device=dev)
    pad[: loc.shape[0]] = loc.reshape(-1, 3)
    parts = [torch.zeros((m, 3), dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    rows = torch.cat([p[: int(s.item())] for p, s in zip(parts, sizes)]).cpu().numpy()
    return sum_triples(rows)


def sum_triples(rows):
    """uint32[n, 3] with duplicate (barcode, conreci) keys summed, sorted by key"""
    rows = np.asarray(rows, dtype=np.int64).reshape(-1, 3)
    if len(rows) == 0:
        return np.zeros((0, 3), dtype=np.uint32)
    key = rows[:, 0] * (1 << 32) + rows[:, 1]
    uk, inv = np.unique(key, return_inverse=True)
    cnt = np.bincount(inv, weights=rows[:, 2]).astype(np.int64)
    return np.stack([uk >> 32, uk & 0xFFFFFFFF, cnt], axis=1).astype(np.uint32)


def sum_stats(stats, group=None):
    """element-wise sum of the 8 arks_map_stats counters over ranks"""
    import torch
    import torch.distributed as dist
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    t = torch.as_tensor(np.asarray(stats, dtype=np.int64)).to(dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().numpy()


# ---- sharded index ------------------------------------------------------------------------------

def pack_vote(count, conreci):
    """the 64-bit vote of arks_map_votes_device as a Python int"""
    return ((int(count) << 32) | (~int(conreci) & 0xFFFFFFFF)) if count > 0 else 0


def unpack_vote(v):
    """(count, conreci) of one vote"""
    v = int(v)
    return (v >> 32, (~v) & 0xFFFFFFFF) if (v >> 32) > 0 else (0, 0)


def reduce_votes(votes, group=None):
    """in-place all-reduce(MAX) of the per-read votes (int64 tensor; on the GPU with RCCL, on the
    CPU with gloo): the vote of the end that wins bestContig's walk over the whole index"""
    import torch.distributed as dist
    if votes.is_cuda and dist.get_backend(group) != "nccl":
        # gloo (ranks sharing one GPU in the tests): through host memory
        host = votes.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.MAX, group=group)
        votes.copy_(host)
    else:
        dist.all_reduce(votes, op=dist.ReduceOp.MAX, group=group)
    return votes


def map_pairs_sharded(shard_index, reads, j_index, pair_ok=None, barcode_id=None, imap=None,
                      stored=None, group=None):
    """The per-pair flow of chromiumRead (Arcs/Arcs.cpp:1264-1292) with the index sharded over the
    ranks of `group`: every rank calls this with the same resident batch and its own shard; all get
    the same (conreci, pair) tensors.  Only one rank should pass `imap` (or the triples of all ranks
    must not be summed): each rank would record every stored pair."""
    from . import api
    ev = api.pair_gate(reads, pair_ok)
    votes = api.map_votes_packed(shard_index, reads, eval_mask=ev)
    reduce_votes(votes, group)
    conreci = api.resolve_votes(votes, reads, shard_index.k, j_index)
    pair = api.pairs_rule(conreci, reads, pair_ok, barcode_id, imap, stored)
    return conreci, pair


class ShardedPairStep:
    """map_pairs_sharded with every buffer preallocated (bench.py --sharded-index)"""

    def __init__(self, shard_index, reads, j_index, pair_ok=None, barcode_id=None, imap=None, group=None):
        import torch
        dev = reads.codes.device
        self.index, self.reads, self.j, self.group = shard_index, reads, float(j_index), group
        self.pair_ok, self.barcode_id, self.imap = pair_ok, barcode_id, imap
        self.n_pairs = reads.n_reads // 2
        self.votes = torch.empty(max(reads.n_reads, 1), dtype=torch.int64, device=dev)
        self.conreci = torch.empty(max(reads.n_reads, 1), dtype=torch.int32, device=dev)
        self.pair = None

    def run(self, stats=None, stored=None, map_events=None):
        import torch.distributed as dist
        from . import api
        assert stats is None, "the window counters of a shard are not the reference's"
        ev = api.pair_gate(self.reads, self.pair_ok)
        if map_events is not None:
            map_events[0].record()
        api.map_votes_packed(self.index, self.reads, eval_mask=ev, out=self.votes)
        if map_events is not None:
            map_events[1].record()
        if dist.is_initialized() and dist.get_world_size(self.group) > 1:
            reduce_votes(self.votes, self.group)
        api.resolve_votes(self.votes, self.reads, self.index.k, self.j, out=self.conreci)
        self.pair = api.pairs_rule(self.conreci, self.reads, self.pair_ok, self.barcode_id, self.imap, stored)


# ---- sharded seed table (BASELINE configs[3]) ------------------------------------------------------------

def _all_to_all(out, inp, out_splits, in_splits, group=None):
    """all_to_all_single; with gloo and device tensors (ranks sharing one GPU in the tests) through host memory"""
    import torch.distributed as dist
    if inp.is_cuda and dist.get_backend(group) != "nccl":
        h_out = out.cpu()
        dist.all_to_all_single(h_out, inp.cpu(), out_splits, in_splits, group=group)
        out.copy_(h_out)
    else:
        dist.all_to_all_single(out, inp, out_splits, in_splits, group=group)
    return out


def exchange_seeds(index, mmer, owner, group=None):
    """The north star's all-to-all: every seed (an 8-byte canonical m-mer) goes to the rank that owns it, the
    owner looks it up in its shard of the seed table, the 16-byte answers come back in the seeds' order.
    Returns int64[2 n_seeds]."""
    import torch
    import torch.distributed as dist
    from . import api
    import os
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not os.environ.get("ARKS_FORCE_EXCHANGE")):
        return api.seeds_probe(index, mmer.contiguous())               # one rank owns every seed
    world = dist.get_world_size(group)
    dev = mmer.device
    n = int(mmer.numel())
    # seeds grouped by owner: a stable sort of one-byte keys (35 ms for 1e8 seeds as an int64 argsort, 1.7 ms so),
    # the group sizes from the sorted keys
    sorted_owner, order = torch.sort(owner.to(torch.uint8), stable=True)
    send = mmer[order].contiguous()
    assert world < 255
    bounds = torch.searchsorted(sorted_owner, torch.arange(world + 1, device=dev, dtype=torch.uint8))
    send_counts = (bounds[1:] - bounds[:-1]).to(torch.int64)
    recv_counts = torch.empty(world, dtype=torch.int64, device=dev)
    _all_to_all(recv_counts, send_counts, None, None, group)
    sc, rc = send_counts.tolist(), recv_counts.tolist()
    asked = torch.empty(max(sum(rc), 1), dtype=torch.int64, device=dev)[:sum(rc)]
    _all_to_all(asked, send, rc, sc, group)
    answers = api.seeds_probe(index, asked)                            # the owner's part
    back = torch.empty(max(2 * n, 2), dtype=torch.int64, device=dev)[:2 * n]
    _all_to_all(back, answers.contiguous(), [2 * x for x in sc], [2 * x for x in rc], group)
    out = torch.empty_like(back)
    out.view(-1, 2)[order] = back.view(-1, 2)                          # back to the seeds' own order
    return out


def map_reads_seed_sharded(index, reads, j_index, eval_mask=None, stats=None, group=None):
    """bestContig of this rank's reads against a seed table sharded over the ranks of `group`
    (ArksIndex.build_seed_shard on every rank).  Every rank must call it in step (the exchange is collective),
    each with its own reads; ranks may hold different numbers of reads, also none."""
    import torch
    from . import api
    counts = api.seed_counts(index, reads, eval_mask)
    seed_off = torch.zeros(reads.n_reads + 1, dtype=torch.int64, device=reads.codes.device)
    seed_off[1:] = torch.cumsum(counts.to(torch.int64), 0)
    mmer, owner = api.seeds_fill(index, reads, seed_off, eval_mask)
    answers = exchange_seeds(index, mmer, owner, group)
    return api.map_reads_seeded(index, reads, j_index, seed_off, answers, eval_mask=eval_mask, stats=stats)


def map_pairs_seed_sharded(index, reads, j_index, pair_ok=None, barcode_id=None, imap=None, stored=None,
                           stats=None, group=None):
    """chromiumRead's per-pair flow (Arcs.cpp:1264-1292) for this rank's read pairs, seed table sharded"""
    from . import api
    ev = api.pair_gate(reads, pair_ok)
    conreci = map_reads_seed_sharded(index, reads, j_index, eval_mask=ev, stats=stats, group=group)
    pair = api.pairs_rule(conreci, reads, pair_ok, barcode_id, imap, stored)
    return conreci, pair
