import sys, os, time, json
sys.path.insert(0, os.getcwd())
import torch, bench, arcs_amd
from arcs_amd import synth
dev = torch.device("cuda", 0)
log = lambda m: print("[ab_gate]", m, file=sys.stderr, flush=True)
wl = bench.Workload(3000.0, 500_000_000, 250_000_000, 60, 0.55, dev, 0, log, want_stats=False)
def timed(fused, steps=6):
    for s in wl.steps: s.fused = fused
    wl.run(); torch.cuda.synchronize()
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in wl.steps] for _ in range(steps)]
    t0 = time.perf_counter()
    for s in range(steps): wl.run(events=ev[s])
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps * 1e3
    km = sum(a.elapsed_time(b) for row in ev for a, b in row) / (steps * len(wl.steps))
    return el, km
for rnd in range(3):
    for fused in (True, False):
        el, km = timed(fused)
        print(f"round {rnd} fused={fused}: {el:.2f} ms per step, map call {km:.2f} ms per launch (events around the map call{' incl. the gate' if fused else ''})", flush=True)
