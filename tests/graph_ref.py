"""Python restatement (test infrastructure) of the stages after the read mapping: IndexMap -> PairMap
-> scaffold graph -> output files (Arcs/Arcs.cpp:833-861, 1378-1526, 1549-1757; Arcs/Arcs.h:185-229;
Graph/DotIO.h:82-114) and of the -D distance estimates (Arcs/DistanceEst.h, Common/MapUtil.h,
Common/StatUtil.h).  numpy float32 reproduces the reference's float arithmetic."""
import bisect
import math

import numpy as np

F = np.float32


def normal_estimation(x, p, n):
    """Arcs.cpp:833-839: float mean and sd, double division / erf, float result"""
    p = F(p)
    mean = F(F(n) * p)
    sd = F(np.sqrt(F(F(F(n) * p) * F(F(1) - p))))
    if sd == 0:
        num = float(F(F(x) - mean))
        z = math.copysign(math.inf, num) if num != 0 else math.nan
    else:
        z = float(F(F(x) - mean)) / (float(sd) * math.sqrt(2.0))
    return F(0.5 * (1 + math.erf(z))) if not math.isnan(z) else F(math.nan)


def head_or_tail(head, tail, P):
    mx, s = max(head, tail), head + tail
    if s < P["min_reads"]:
        return (False, False)
    cdf = normal_estimation(mx, 0.5, s)
    if F(F(1) - cdf) < F(P["error_percent"]):
        return (True, mx == head)
    return (False, False)


def add_opposite_ends(imap):
    for bc, sm in imap.items():
        for (ctg, h) in list(sm):
            sm.setdefault((ctg, not h), 0)


def pair_contigs(imap, mult, P):
    pmap = {}
    for bc, sm in imap.items():
        m = mult.get(bc, 0)
        if not (P["min_mult"] <= m <= P["max_mult"]):
            continue
        heads = sorted(c for (c, h) in sm if h)
        for a in heads:
            for b in heads:
                if not a < b:
                    continue
                va = head_or_tail(sm.get((a, True), 0), sm.get((a, False), 0), P)
                vb = head_or_tail(sm.get((b, True), 0), sm.get((b, False), 0), P)
                if va[0] and vb[0]:
                    cnt = pmap.setdefault((a, b), [0, 0, 0, 0])
                    cnt[(0 if va[1] else 2) + (0 if vb[1] else 1)] += 1
    return dict(sorted(pmap.items()))


def create_graph(pmap, P):
    ids, edges, vmap = [], [], {}
    for (a, b), count in pmap.items():
        mx, index = 0, 0
        for i, c in enumerate(count):
            if c > mx:
                mx, index = c, i
        second = max([c for c in count if c != mx] + [0])
        if mx < P["min_links"]:
            continue
        cdf = normal_estimation(mx, 0.5, mx + second)
        if not (F(F(1) - cdf) < F(P["error_percent"])):
            continue
        for s in (a, b):
            if s not in vmap:
                vmap[s] = len(ids)
                ids.append(s)
        edges.append((vmap[a], vmap[b], index, mx))
    return ids, edges


def remove_degree_nodes(ids, edges, max_degree):
    deg = [0] * len(ids)
    for (u, v, _, _) in edges:
        deg[u] += 1
        deg[v] += 1
    dead = {v for v in range(len(ids)) if deg[v] > max_degree}
    edges = [e for e in edges if e[0] not in dead and e[1] not in dead]
    return dead, edges


def graph_text(ids, edges, dead=()):
    index, n = {}, 0
    for v in range(len(ids)):
        if v not in dead:
            index[v] = n
            n += 1
    out = ["graph G {"]
    out += [f"{index[v]} [id={ids[v]}];" for v in range(len(ids)) if v not in dead]
    out += [f"{index[u]}--{index[v]} [label={o}, weight={w}];" for (u, v, o, w) in edges]
    out.append("}")
    return "\n".join(out) + "\n"


_UMAP_EXE = None


def unordered_map_order(keys):
    """the keys in the iteration order of a std::unordered_map<std::string, int> filled in the given order
    (tests/umap_order.cpp, compiled on first use with the local g++: the same libstdc++ as the product)"""
    global _UMAP_EXE
    import os, subprocess, tempfile
    if _UMAP_EXE is None:
        d = tempfile.mkdtemp(prefix="umap_order_")
        _UMAP_EXE = os.path.join(d, "umap_order")
        subprocess.check_call(["g++", "-O1", "-std=c++17", os.path.join(os.path.dirname(os.path.abspath(__file__)), "umap_order.cpp"),
                               "-o", _UMAP_EXE])
    out = subprocess.run([_UMAP_EXE], input="".join(k + "\n" for k in keys), capture_output=True, text=True, check=True).stdout
    return out.split("\n")[:-1]


def dist_graph_text(lengths, ids, edges, gap, dists=None):
    """<base>.dist.gv as createAbyssGraph + write_dot write it (Arcs.cpp:1615-1659, Graph/DotIO.h:82-114):
    two vertices per contig in the iteration order of the ContigToLength unordered_map (`lengths`: a dict in
    FASTA order), every edge with its reverse-complement twin, out-edges per vertex in insertion order"""
    names = unordered_map_order(list(lengths))
    index = {n: i for i, n in enumerate(names)}
    adj = [[] for _ in range(2 * len(names))]
    vname = lambda v: names[v >> 1] + ("-" if v & 1 else "+")
    for ei, (u, v, o, w) in enumerate(edges):
        a = 2 * index[ids[u]] + (1 if o < 2 else 0)
        b = 2 * index[ids[v]] + (o % 2)
        d = gap if dists is None else dists[ei]
        assert all(x[0] != b for x in adj[a]), "duplicate edge"
        adj[a].append((b, w, d))
        if a != (b ^ 1):
            if all(x[0] != (a ^ 1) for x in adj[b ^ 1]):
                adj[b ^ 1].append((a ^ 1, w, d))
    out = ["digraph arcs {"]
    for v in range(len(adj)):
        out.append(f'"{vname(v)}" [l={lengths[names[v >> 1]]}]')
    for u in range(len(adj)):
        for (v, w, d) in adj[u]:
            out.append(f'"{vname(u)}" -> "{vname(v)}" [d={d} e={float(gap):.1f} n={w}]')
    out.append("}")
    return "\n".join(out) + "\n"


def dist_graph_lines(lengths, ids, edges, gap):
    """(vertex lines, edge lines) as sets: the vertex order is libstdc++'s unordered_map order"""
    vl = set()
    for c, L in lengths.items():
        vl.add(f'"{c}+" [l={L}]')
        vl.add(f'"{c}-" [l={L}]')
    el = set()
    for (u, v, o, w) in edges:
        un = ids[u] + ("-" if o < 2 else "+")
        vn = ids[v] + ("-" if o % 2 else "+")
        el.add(f'"{un}" -> "{vn}" [d={gap} e={float(gap):.1f} n={w}]')
        flip = lambda s: s[:-1] + ("+" if s[-1] == "-" else "-")
        if un != flip(vn):
            el.add(f'"{flip(vn)}" -> "{flip(un)}" [d={gap} e={float(gap):.1f} n={w}]')
    return vl, el


def tsv_text(imap, pmap, mult, P):
    barcode_count = sum(1 for m in mult.values() if P["min_mult"] <= m <= P["max_mult"])
    per_end = {}
    for sm in imap.values():
        for key, c in sm.items():
            if c >= P["min_reads"]:
                per_end[key] = per_end.get(key, 0) + 1
    out = ["U\tV\tBest_orientation\tShared_barcodes\tU_barcodes\tV_barcodes\tAll_barcodes"]
    for (u, v), counts in pmap.items():
        mx = max(counts)
        for i, c in enumerate(counts):
            if c == 0:
                continue
            usense, vsense = i < 2, bool(i % 2)
            best = "T" if c == mx else "F"
            ub, vb = per_end.get((u, usense), 0), per_end.get((v, not vsense), 0)
            out.append(f"{u}{'-' if usense else '+'}\t{v}{'-' if vsense else '+'}\t{best}\t{c}\t{ub}\t{vb}\t{barcode_count}")
            out.append(f"{v}{'+' if vsense else '-'}\t{u}{'+' if usense else '-'}\t{best}\t{c}\t{vb}\t{ub}\t{barcode_count}")
    return "\n".join(out) + "\n"


def pair_text(pmap):
    return "".join(f"{a}\t{b}\t{c[0]}\t{c[1]}\t{c[2]}\t{c[3]}\n" for (a, b), c in pmap.items())


def counts_text(mult):
    rows = sorted(mult.items(), key=lambda x: (-x[1], x[0]))
    return "".join(f"{b}\t{m}\n" for b, m in rows)


# ---- -D: distance estimates (Arcs/DistanceEst.h) ------------------------------------------------

def dist_samples(imap, lengths, mult, P):
    """DistanceEst.h:101-173 -> {contig: [distance, barcodes_head, barcodes_tail, union, intersect]}"""
    out = {}
    for bc, sm in imap.items():
        if not (P["min_mult"] <= mult[bc] <= P["max_mult"]):
            continue
        for (ctg, head), pairs in sm.items():
            if pairs < P["min_reads"] or lengths[ctg] < 2 * P["end_length"]:
                continue
            s = out.setdefault(ctg, [0, 0, 0, 0, 0])
            s[0] = lengths[ctg] - 2 * P["end_length"]
            s[1 if head else 2] += 1
            found_other = sm.get((ctg, not head), -1) >= P["min_reads"] and (ctg, not head) in sm
            if found_other and head:
                s[4] += 1
                s[3] += 1
            elif not found_other:
                s[3] += 1
    return dict(sorted(out.items()))


def jaccard_to_dist(samples):
    """DistanceEst.h:181-189 with this build's tie rule: of equal Jaccard indices the sample of the
    smallest contig id stays (the reference keeps whichever its unordered_map yields first)"""
    j2d = {}
    for ctg, s in samples.items():           # sorted by contig id
        j2d.setdefault(s[4] / s[3], s[0])
    keys = sorted(j2d)
    return keys, [j2d[k] for k in keys]


def closest_key(keys, key):
    """Common/MapUtil.h:8-44 -> position"""
    it = bisect.bisect_left(keys, key)
    if it == 0:
        return 0
    if it == len(keys):
        return it - 1
    return it if abs(key - keys[it - 1]) > abs(key - keys[it]) else it - 1


def closest_keys(keys, key, n):
    """Common/MapUtil.h:47-93 -> [first, last)"""
    if not keys:
        return 0, 0
    first = closest_key(keys, key)
    last = first + 1
    count = 1
    while count < n:
        if first == 0 and last == len(keys):
            break
        if first == 0:
            last += 1
        elif last == len(keys):
            first -= 1
        elif abs(key - keys[first - 1]) < abs(key - keys[last]):
            first -= 1
        else:
            last += 1
        count += 1
    return first, last


def quantile(sorted_vals, q):
    """Common/StatUtil.h:8-32 (the reference's weights: the fraction goes to the LOWER element)"""
    last = len(sorted_vals) - 1
    bpos, apos = math.floor(q * last), math.ceil(q * last)
    w = q * last - bpos
    return w * sorted_vals[bpos] + (1.0 - w) * sorted_vals[apos]


def pair_barcode_stats(imap, mult, lengths, P):
    """DistanceEst.h:220-334 -> {(id1, id2): [[barcodes1, barcodes2, union, intersect] x HH,HT,TH,TT]}"""
    def valid(ctg, pairs):
        return pairs >= P["min_reads"] and lengths[ctg] >= 2 * P["end_length"]
    per_end, out = {}, {}
    for bc, sm in imap.items():
        if not (P["min_mult"] <= mult[bc] <= P["max_mult"]):
            continue
        for (c1, h1), p1 in sm.items():
            if not valid(c1, p1):
                continue
            per_end[(c1, h1)] = per_end.get((c1, h1), 0) + 1
            for (c2, h2), p2 in sm.items():
                if not valid(c2, p2) or c1 > c2:
                    continue
                st = out.setdefault((c1, c2), [[0, 0, 0, 0] for _ in range(4)])
                st[(0 if h1 else 2) + (0 if h2 else 1)][3] += 1
    for (c1, c2), arr in out.items():
        for o in range(4):
            n1 = per_end.get((c1, o < 2))
            if n1 is None:
                continue
            arr[o][0] = n1
            n2 = per_end.get((c2, o % 2 == 0))
            if n2 is None:
                continue
            arr[o][1] = n2
            arr[o][2] = n1 + n2 - arr[o][3]
    return out


def edge_distances(ids, edges, stats, j2d, P):
    """DistanceEst.h:337-430 -> per edge None or (min_dist, dist, max_dist)"""
    keys, dists = j2d
    out = []
    for (u, v, o, w) in edges:
        st = stats.get((ids[u], ids[v]))
        if not keys or st is None or st[o][2] == 0:
            out.append(None)
            continue
        first, last = closest_keys(keys, st[o][3] / st[o][2], P["dist_bin_size"])
        d = sorted(dists[first:last])
        # C's round(): half away from zero (values are non-negative here)
        out.append((math.floor(quantile(d, 0.01)), int(math.floor(quantile(d, 0.5) + 0.5)),
                    math.ceil(quantile(d, 0.99))))
    return out


def graph_text_with_distances(ids, edges, est, dead=()):
    index, n = {}, 0
    for v in range(len(ids)):
        if v not in dead:
            index[v] = n
            n += 1
    out = ["graph G {"]
    out += [f"{index[v]} [id={ids[v]}];" for v in range(len(ids)) if v not in dead]
    for (u, v, o, w), e in zip(edges, est):
        if u in dead or v in dead:
            continue
        extra = f", d={e[1]}, maxd={e[2]}" if e is not None else ""
        out.append(f"{index[u]}--{index[v]} [label={o}, weight={w}{extra}];")
    out.append("}")
    return "\n".join(out) + "\n"


def dist_tsv_text(ids, edges, est, stats, dead=()):
    out = ["contig1\tcontig2\tmin_dist\tdist\tmax_dist\tbarcodes1\tbarcodes2\tbarcodes_union\tbarcodes_intersect"]
    for (u, v, o, w), e in zip(edges, est):
        if u in dead or v in dead:
            continue
        st = stats.get((ids[u], ids[v]))
        if st is None:
            continue
        s1, s2 = o < 2, bool(o % 2)
        d = "\t".join(map(str, e)) if e is not None else "NA\tNA\tNA"
        b1, b2, un, it = st[o]
        out.append(f"{ids[u]}{'-' if s1 else '+'}\t{ids[v]}{'-' if s2 else '+'}\t{d}\t{b1}\t{b2}\t{un}\t{it}")
        out.append(f"{ids[v]}{'+' if s2 else '-'}\t{ids[u]}{'+' if s1 else '-'}\t{d}\t{b2}\t{b1}\t{un}\t{it}")
    return "\n".join(out) + "\n"


def samples_text(samples):
    out = ["contig_id\tdistance\tbarcodes_head\tbarcodes_tail\tbarcodes_union\tbarcodes_intersect"]
    out += [f"{c}\t{s[0]}\t{s[1]}\t{s[2]}\t{s[3]}\t{s[4]}" for c, s in samples.items()]
    return "\n".join(out) + "\n"
