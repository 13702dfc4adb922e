"""usage: python profiles/tools/trace_busy.py <rocprofv3 *_kernel_trace.csv> [t0_frac t1_frac]
Where the wall time of a step goes when many kernels of several streams overlap (the sharded seed table on one GPU: 8
local ranks x 2 batches in flight): the span of the trace (or of the [t0_frac, t1_frac) part of it), the time at
least one kernel runs, and per kernel name (templates cut) the summed durations AND its share of the busy time with
every instant divided evenly among the kernels that run in it -- the shares add up to the busy time."""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", re.sub(r"^void ", "", r["Kernel_Name"]))) for r in rows]
ks.sort()
lo, hi = ks[0][0], max(k[1] for k in ks)
if len(sys.argv) > 3:
    a, b = float(sys.argv[2]), float(sys.argv[3])
    lo, hi = lo + int((hi - lo) * a), lo + int((hi - lo) * b)
ev = []
for s, e, n in ks:
    s, e = max(s, lo), min(e, hi)
    if e > s:
        ev.append((s, 1, n)); ev.append((e, -1, n))
ev.sort(key=lambda t: (t[0], t[1]))
active = collections.Counter(); share = collections.Counter(); total = collections.Counter(); calls = collections.Counter()
busy = 0; last = lo; nact = 0; conc = collections.Counter()
for t, d, n in ev:
    if nact > 0 and t > last:
        dt = t - last; busy += dt; conc[min(nact, 16)] += dt
        for name, c in active.items():
            if c: share[name] += dt * c / nact
    last = t
    active[n] += d; nact += d
for s, e, n in ks:
    s2, e2 = max(s, lo), min(e, hi)
    if e2 > s2: total[n] += e2 - s2; calls[n] += 1
span = hi - lo
print(f"span {span / 1e6:.2f} ms, a kernel running {busy / 1e6:.2f} ms ({100.0 * busy / span:.1f} %), idle {(span - busy) / 1e6:.2f} ms; kernels {sum(calls.values())}")
print("concurrency (kernels running : ms): " + " ".join(f"{c}:{v / 1e6:.1f}" for c, v in sorted(conc.items())))
print(f"{'kernel':58s} {'calls':>6s} {'sum ms':>9s} {'share ms':>9s} {'share %':>8s}")
for n, v in share.most_common(14):
    print(f"{n[:58]:58s} {calls[n]:6d} {total[n] / 1e6:9.2f} {v / 1e6:9.2f} {100.0 * v / max(busy, 1):8.1f}")
