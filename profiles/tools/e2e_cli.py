"""End-to-end throughput of the `arcs --arks` CLI: FASTQ(.gz) files -> .gv, by -t."""
import gzip, os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from arcs_amd import build as b, synth
exe = b.build_host()
tmp = "/tmp/e2e"; os.makedirs(tmp, exist_ok=True)
NF, NP = int(os.environ.get("E2E_FILES", 4)), int(os.environ.get("E2E_PAIRS", 1000000))
DRAFT_MBP = float(os.environ.get("E2E_DRAFT_MBP", 20))     # 3000 = a human-size draft (BASELINE configs[2])
contigs = synth.make_draft(int(DRAFT_MBP * 1e6), seed=7)
with open(f"{tmp}/draft.fa", "wb") as f:
    for i, c in enumerate(contigs):
        f.write(b">%d\n" % (i + 1)); f.write(c.tobytes()); f.write(b"\n")
print("draft: %d contigs, %.0f Mbp" % (len(contigs), sum(len(c) for c in contigs) / 1e6), flush=True)
mult = {}
t0 = time.time()
import zlib
from concurrent.futures import ThreadPoolExecutor

def one_stream_gz(args):            # a real single-stream .gz (what a sequencer's pipeline writes), level 1
    path, data = args
    c = zlib.compressobj(1, zlib.DEFLATED, 31)
    with open(path, "wb") as f:
        mv = memoryview(data)
        for i in range(0, len(mv), 64 << 20):
            f.write(c.compress(mv[i:i + (64 << 20)]))
        f.write(c.flush())

jobs = []
for fi in range(NF):
    batch = synth.make_read_pairs(contigs, NP, seed=100 + fi, device="cuda" if DRAFT_MBP > 500 else "cpu")
    batch = {k: v.cpu() for k, v in batch.items()}
    batch["barcode_id"] = batch["barcode_id"] + fi * (NP // 80 + 1)          # barcodes of their own per file
    text = synth.fastq_bytes(batch, first_pair=fi * NP).tobytes()
    for b in np.unique(batch["barcode_id"].numpy()):
        v, name = int(b), []
        for _ in range(16):
            name.append("ACGT"[v % 4]); v //= 4
        mult["".join(reversed(name)) + "-1"] = 160
    open(f"{tmp}/r{fi}.fq", "wb").write(text)
    jobs.append((f"{tmp}/r{fi}.fq.gz", text))
    if os.environ.get("E2E_BGZF"):
        sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
        from test_host_ingest import write_bgzf
        write_bgzf(f"{tmp}/r{fi}.bgzf.fq.gz", text, level=4)
with ThreadPoolExecutor(32) as ex:
    list(ex.map(one_stream_gz, jobs))
del jobs
with open(f"{tmp}/mult.tsv", "w") as f:
    f.writelines(f"{k}\t{v}\n" for k, v in mult.items())
print("data generated in %.1f s; %d files x %d pairs; fq %.0f MB each" % (time.time()-t0, NF, NP, len(text)/1e6), flush=True)
ref = None
EXTS = os.environ.get("E2E_EXTS")
PREFIX = os.environ.get("E2E_PREFIX", "").split()
for ext in (EXTS.split(",") if EXTS else (".fq", ".fq.gz", ".bgzf.fq.gz") if os.environ.get("E2E_BGZF") else (".fq", ".fq.gz")):
    files = [f"{tmp}/r{fi}{ext}" for fi in range(NF)] * int(os.environ.get("E2E_REPEAT", 1))
    for t in [int(x) for x in os.environ.get('E2E_THREADS', '1,4,16').split(',')]:
        args = PREFIX + [exe, "--arks", "-f", f"{tmp}/draft.fa", "-u", f"{tmp}/mult.tsv", "-k", "60", "-j", "0.55", "-c", "5",
                "-m", "50-10000", "-e", "30000", "-z", "500", "-t", str(t), "-b", f"{tmp}/out_{t}{ext.replace('.', '_')}"] + \
            os.environ.get("E2E_ARGS", "").split() + files
        t1 = time.time()
        out = subprocess.run(args, capture_output=True, text=True, env=dict(os.environ, ARKS_TIMING='1'))
        dt = time.time() - t1
        if out.returncode:
            print(out.stdout[-2000:], out.stderr[-2000:]); sys.exit(1)
        gv = open(f"{tmp}/out_{t}{ext.replace('.', '_')}_original.gv").read()
        ref = ref or gv
        print(f"{ext:7s} -t {t:2d}: {dt:6.2f} s total  ({len(files)*NP/dt/1e6:.2f} M pairs/s whole run)  gv_same={gv == ref}", flush=True)
        rd = [l for l in out.stderr.splitlines() if 'read files' in l]
        ms = float(rd[0].split(':')[1].split()[0]) if rd else 0
        print('        reads stage %.0f ms -> %.2f M pairs/s' % (ms, len(files)*NP/ms/1e3), flush=True)
        for l in out.stderr.splitlines():
            if l.startswith('ingest profile') or (os.environ.get('E2E_TIMING') and l.startswith(('[timing]', 'pgzip:'))):
                print('        ' + l)
