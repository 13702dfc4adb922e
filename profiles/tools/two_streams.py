"""Does a second stream buy anything?  The launches of a step alternate between two streams (what the CLI's lanes do
with their two buffer sets): launch i + 1's hot kernel may start while launch i's medium kernel and pair rule finish.
usage: two_streams.py [human] [pairs] [chunk]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import bench
human = "human" in sys.argv[1:]
nums = [int(a) for a in sys.argv[1:] if a.isdigit()]
pairs = nums[0] if nums else 250_000_000
chunk = nums[1] if len(nums) > 1 else pairs // 2
dev = torch.device("cuda", 0)
log = lambda m: print("[two_streams]", m, file=sys.stderr, flush=True)
wl = bench.Workload(3000.0, pairs, chunk, 60, 0.55, dev, 0, log, want_stats=False, repeats="human" if human else None)
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]


def seq():
    for s in wl.steps:
        s.run()


def alt():
    for i, s in enumerate(wl.steps):
        with torch.cuda.stream(streams[i % 2]):
            s.run()


for name, fn in (("one stream", seq), ("two streams", alt), ("one stream", seq), ("two streams", alt)):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 4
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n
    print(f"{'human-like' if human else 'uniform'} draft, {pairs} pairs in {len(wl.steps)} launches, {name}: {ms:.2f} ms per pass", flush=True)
t = wl.imap.triples()
print("imap entries", len(t), "count sum", int(t[:, 2].sum()))
