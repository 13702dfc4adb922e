"""(run from the repository root) interleaved A/B timing of arks_map_reads_device across several builds of libarks_hip.so
usage: python scratch/ab.py name=path.so [name=path.so ...]"""
import ctypes as C, os, sys, statistics
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from arcs_amd import synth, _lib
import arcs_amd
from arcs_amd.api import _concat

variants = [a.split("=", 1) for a in sys.argv[1:]]
k, j = 60, 0.55
NP = int(os.environ.get("AB_PAIRS", 4_000_000))
contigs = synth.make_draft(int(float(os.environ.get('AB_DRAFT_MBP', 50)) * 1e6), seed=synth.SEED,
                           repeats={"": False, "human": "human", "1": True}[os.environ.get("AB_REPEATS", "")])
cs = synth.contigs_to_strings(contigs)
ends = arcs_amd.contig_ends(cs) if False else None
# contig ends without touching the default lib
def cutoff(L, mn=500, e=30000):
    if L < mn: return None
    c = e
    if c == 0 or L <= 2 * c: c = L // 2
    return c
ends = []
for s_ in cs:
    c = cutoff(len(s_))
    if c is None: continue
    ends.append(s_[:c]); ends.append(s_[len(s_) - c:])
data, offsets, lens = _concat(ends)
data = np.concatenate([data, np.zeros(1, np.uint8)])
batch = synth.make_read_pairs(contigs, NP, seed=synth.SEED + 1, device="cuda")
n = int(batch["lens"].numel())
dev = torch.device("cuda", 0)
woff = torch.zeros(n + 1, dtype=torch.int64, device=dev)
woff[1:] = torch.cumsum((batch["lens"].to(torch.int64) + 31) // 32, 0)
total = int(woff[-1].item())
d_ascii = torch.cat([batch["ascii"], torch.zeros(64, dtype=torch.uint8, device=dev)])
sp = C.c_void_p(torch.cuda.current_stream(0).cuda_stream)
libs = []
for name, path in variants:
    L = C.CDLL(os.path.abspath(path))
    L.arks_index_build.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
    L.arks_pack_reads_device.argtypes = [C.c_void_p] * 4 + [C.c_int64] + [C.c_void_p] * 3 + [C.c_int, C.c_void_p]
    L.arks_map_reads_device.argtypes = [C.c_void_p] * 6 + [C.c_int64, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
    h = C.c_void_p()
    assert L.arks_index_build(C.byref(h), k, data.ctypes.data, offsets.ctypes.data, lens.ctypes.data, len(lens), 0, None) == 0
    codes = torch.zeros(total + 4, dtype=torch.int64, device=dev)
    nmask = torch.zeros(total + 4, dtype=torch.int32, device=dev)
    rclass = torch.zeros(n, dtype=torch.uint8, device=dev)
    assert L.arks_pack_reads_device(d_ascii.data_ptr(), batch["offsets"].data_ptr(), batch["lens"].data_ptr(), woff.data_ptr(), n, codes.data_ptr(), nmask.data_ptr(), rclass.data_ptr(), 0, sp) == 0
    out = torch.zeros(n, dtype=torch.int32, device=dev)
    ev = None
    if os.environ.get("AB_EVAL"):          # the gate's eval array (0 / 1 / 3), as bench.py and the CLI map
        L.arks_pair_gate_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p]
        ev = torch.zeros(n, dtype=torch.uint8, device=dev)
        assert L.arks_pair_gate_device(batch["pair_ok"].data_ptr(), rclass.data_ptr(), n // 2, ev.data_ptr(), 0, sp) == 0
    libs.append((name, L, h, codes, nmask, out, ev))
torch.cuda.synchronize()
d_stats = torch.zeros(8, dtype=torch.int64, device=dev)
def run(L, h, codes, nmask, out, ev=None, name=""):
    st = d_stats.data_ptr() if (os.environ.get("AB_STATS") or name.endswith("+stats")) else None
    assert L.arks_map_reads_device(h, codes.data_ptr(), nmask.data_ptr(), woff.data_ptr(), batch["lens"].data_ptr(), ev.data_ptr() if ev is not None else None, n, j, out.data_ptr(), st, sp) == 0
for (name, L, h, codes, nmask, out, ev) in libs:
    run(L, h, codes, nmask, out, ev)
torch.cuda.synchronize()
ref = libs[0][5].clone()
times = {name: [] for name, *_ in libs}
for rnd in range(7):
    for (name, L, h, codes, nmask, out, ev) in libs:
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); run(L, h, codes, nmask, out, ev, name); b.record(); torch.cuda.synchronize()
        times[name].append(a.elapsed_time(b))
windows = int(torch.clamp(batch["lens"].to(torch.int64) - (k - 1), min=0).sum().item())
# the read stream of one launch, byte for byte (profiles/tools/fetch_calib.py: a build without probes fetches nothing else)
_ev = libs[0][6]
_n_eval = int((_ev != 0).sum().item()) if _ev is not None else n
_may_n = int((_ev == 1).sum().item()) if _ev is not None else n
print(f"stream_bytes codes={8 * total} nmask_if_all={4 * total} word_off={8 * (n + 1)} lens={4 * n} eval={n if _ev is not None else 0} "
      f"out={4 * n} reads={n} evaluated={_n_eval} reads_whose_masks_may_be_fetched={_may_n}")
for (name, L, h, codes, nmask, out, ev) in libs:
    t = times[name]
    same = bool((out == ref).all().item())
    try:
        fs = (C.c_int64 * 2)(); L.arks_index_fallback_size.argtypes = [C.c_void_p, C.c_void_p]
        if L.arks_index_fallback_size(h, fs) == 0: print("   exact table behind heavy seeds:", fs[0], "keys,", fs[1], "bytes")
    except AttributeError:
        pass
    try:
        qc = (C.c_uint * 4)(); L.arks_debug_queue_counts.argtypes = [C.c_void_p, C.c_void_p]; L.arks_debug_queue_counts(h, qc); print("   queues: slow", qc[0], "medium", qc[2], "of", n, "reads")
    except AttributeError:
        pass
    try:
        sec = (C.c_ulonglong * 16)(); L.arks_debug_section_cycles.argtypes = [C.c_void_p]; L.arks_debug_section_cycles(sec)
        tot = sum(sec[:12]) or 1
        print("   sections (share of the hot kernel's wave cycles):", " ".join(f"{i}:{100.0 * sec[i] / tot:.1f}" for i in range(12)), f"total {tot:.3e}")
    except AttributeError:
        pass
    try:
        ad = (C.c_ulonglong * 8)(); L.arks_debug_alt_round.argtypes = [C.c_void_p]; L.arks_debug_alt_round(ad)
        print("   alt round (all launches of this process): candidates", ad[0], "finished", ad[1], "| words > 4 differing bases", ad[2],
              "list full", ad[3], "seed with entries", ad[4], "seed heavy / > 2 entries", ad[5], "open window without anchor (words)", ad[6],
              "candidates without diagonal A", ad[7])
    except AttributeError:
        pass
    try:   # a library built with -DARKS_MEDIUM_DIAG (quarantined diagnostic: counters of the medium kernel's tiles)
        md = (C.c_ulonglong * 16)(); L.arks_debug_medium_diag.argtypes = [C.c_void_p]; L.arks_debug_medium_diag(md)
        print("   medium kernel (all launches of this process): tiles", md[0], "reads", md[1], "with a diagonal", md[2], "seeds probed", md[9],
              "tiles with a second diagonal", md[11], "| windows proven absent by a seed without entries", md[6] >> 32,
              "by a seed whose entries are all staged", md[6] & 0xFFFFFFFF, "| windows left to T6d", md[3],
              "(in reads without a diagonal", md[4], ", in reads whose diagonal A differs in > 8 bases", md[5], ")",
              "| reads settled before any lookup", md[13], "with a first round", md[14], "settled after it", md[10],
              "| windows looked up", md[7], "probe rounds", md[8], "slot reads", md[12])
    except AttributeError:
        pass
    dig = int((out.to(torch.int64) * (torch.arange(n, device=dev, dtype=torch.int64) % 1000003 + 1)).sum().item())
    print(f"{name:14s} digest {dig} nonzero {int((out != 0).sum().item())}")
    print(f"{name:14s} median {statistics.median(t):7.3f} ms  min {min(t):7.3f}  -> {windows / (statistics.median(t) * 1e-3) / 1e9:6.2f} G k-mers/s  same_as_first={same}")
