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
for fi in range(NF):
    batch = synth.make_read_pairs(contigs, NP, seed=100 + fi, device="cuda" if DRAFT_MBP > 500 else "cpu")
    batch = {k: v.cpu() for k, v in batch.items()}
    reads = synth.reads_to_strings(batch)
    bid = batch["barcode_id"].numpy()
    parts = []
    for p in range(NP):
        bc = f"BC{int(bid[p]):08d}-1"
        mult[bc] = mult.get(bc, 0) + 2
        s1, s2 = reads[2*p], reads[2*p+1]
        parts.append(f"@r{fi}_{p}/1 BX:Z:{bc}\n{s1}\n+\n{'F'*len(s1)}\n@r{fi}_{p}/2 BX:Z:{bc}\n{s2}\n+\n{'F'*len(s2)}\n")
    text = "".join(parts)
    open(f"{tmp}/r{fi}.fq", "w").write(text)
    with gzip.open(f"{tmp}/r{fi}.fq.gz", "wt", compresslevel=4) as f:
        f.write(text)
    if os.environ.get("E2E_BGZF"):
        sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
        from test_host_ingest import write_bgzf
        write_bgzf(f"{tmp}/r{fi}.bgzf.fq.gz", text.encode(), level=4)
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
                "-m", "50-10000", "-e", "30000", "-z", "500", "-t", str(t), "-b", f"{tmp}/out_{t}{ext.replace('.', '_')}"] + files
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
