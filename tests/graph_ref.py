"""Python restatement (test infrastructure) of the stages after the read mapping: IndexMap -> PairMap
-> scaffold graph -> output files (Arcs/Arcs.cpp:833-861, 1378-1526, 1549-1757; Arcs/Arcs.h:185-229;
Graph/DotIO.h:82-114).  numpy float32 reproduces the reference's float arithmetic."""
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
