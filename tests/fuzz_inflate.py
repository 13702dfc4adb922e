"""Damaged-input fuzz of arcs_amd/host/fast_inflate.hpp and pgzip.hpp under AddressSanitizer + UBSan (not a
pytest test): bit flips, truncations and overwritten spans of gzip files of every block type; every run must end
without a sanitizer report, a crash or a hang, and the stream decoded by several threads (pgzip.hpp, chunks of a
few kilobytes) must be the one-thread inflater's.  usage: python tests/fuzz_inflate.py [cases]
(round 1: 700 cases clean)"""
import gzip, os, random, subprocess, sys, tempfile, zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "arcs_amd", "host")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 700
tmp = tempfile.mkdtemp()
exe = os.path.join(tmp, "inflate_check_asan")
subprocess.check_call(["g++", "-O1", "-g", "-pthread", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-std=c++17",
                       "-I" + HOST, os.path.join(HOST, "inflate_check.cpp"), "-lz", "-ldl", "-o", exe])
random.seed(5)
text = "".join(f"@r{i}\n{''.join(random.choice('ACGTN') for _ in range(random.randint(40, 160)))}\n+\n{'F' * 60}\n"
               for i in range(3000)).encode()
blobs = [gzip.compress(text, lvl) for lvl in (1, 6, 9)]
c = zlib.compressobj(6, zlib.DEFLATED, 31, 8, zlib.Z_FIXED)
blobs.append(c.compress(text[:50000]) + c.flush())
blobs.append(gzip.compress(os.urandom(70000), 6))
bad = ran = 0
for t in range(n_cases):
    b = bytearray(random.choice(blobs))
    mode = random.random()
    if mode < 0.6:
        for _ in range(random.randint(1, 4)):
            b[random.randrange(len(b))] ^= 1 << random.randrange(8)
    elif mode < 0.8:
        b = b[:random.randrange(3, len(b))]
    else:
        p = random.randrange(10, len(b))
        b[p:p + random.randint(1, 40)] = os.urandom(random.randint(1, 40))
    if bytes(b[:3]) != b"\x1f\x8b\x08":
        continue
    path = os.path.join(tmp, "case.gz")
    open(path, "wb").write(bytes(b))
    r = subprocess.run([exe, path, str(1 << 18), "pgz", "3", str(random.choice((3000, 9001, 40000))), "5"],
                       capture_output=True, text=True, timeout=60)
    ran += 1
    if "ERROR" in r.stderr or "runtime error" in r.stderr or r.returncode not in (0, 1) or "pgz same " not in r.stdout:
        bad += 1
        print("case", t, r.returncode, r.stdout[-300:], r.stderr[-800:])
print("cases", ran, "sanitizer failures", bad)
sys.exit(1 if bad else 0)
