"""Builds arcs_amd/lib/libarks_hip.so with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SOURCES = [os.path.join(HERE, "csrc", f) for f in ("arks_build.hip", "arks_map.hip", "arks_shard.hip", "arks_imap.hip", "arks_capi.hip")]
HEADERS = [os.path.join(HERE, "csrc", f) for f in ("arks_device.hpp", "arks_kernels.hpp", "arks_exchange.hpp", "arks_shard_stats.hpp")] + \
          [os.path.join(ROOT, "include", f) for f in ("arks_hip.h", "arks_hip_debug.h")]
OUT = os.path.join(HERE, "lib", "libarks_hip.so")


STAMP = OUT + ".digest"          # digest of the sources the library was built from (travels with it)


def _digest(paths):
    import hashlib
    h = hashlib.sha256()
    for p in paths:
        h.update(os.path.basename(p).encode() + b"\0")
        h.update(open(p, "rb").read())
    return h.hexdigest()


def needs_build():
    """by content, not by time stamp: a copy of the tree (the GPU box gets one) must not look newer than the library
    that came with it -- a rebuild there costs minutes of hipcc inside a test run"""
    if not os.path.exists(OUT):
        return True
    try:
        return open(STAMP).read().strip() != _digest(SOURCES + HEADERS)
    except OSError:
        t = os.path.getmtime(OUT)
        return any(os.path.getmtime(p) > t for p in SOURCES + HEADERS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(HERE, "csrc"),
           *SOURCES, "-o", OUT]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(_digest(SOURCES + HEADERS) + "\n")
    return OUT


HOST_SOURCES = [os.path.join(HERE, "host", f) for f in ("arcs.cpp", "graph.hpp", "dist_est.hpp", "seqio.hpp", "ingest.hpp", "graph_fast.hpp", "rank_merge.hpp", "bgzf.hpp", "pgzip.hpp", "fast_inflate.hpp", "crc32_fold.hpp", "long_to_linked_pe.cpp")]
HOST_OUT = os.path.join(HERE, "bin", "arcs")


def build_host(force=False, verbose=False):
    """the `arcs --arks` front end (C++17 host program over the C ABI; needs zlib)"""
    build(force=False, verbose=verbose)
    host_stamp = HOST_OUT + ".digest"
    if not os.path.exists(STAMP):
        # a library that needs_build() accepted by its time stamps (built by an older build.py, or a copied tree
        # without the stamp): write the stamp it would have had
        with open(STAMP, "w") as f:
            f.write(_digest(SOURCES + HEADERS) + "\n")
    host_digest = _digest(HOST_SOURCES + [os.path.join(ROOT, "include", "arks_hip.h"), STAMP])
    if not force and os.path.exists(HOST_OUT) and os.path.exists(os.path.join(HERE, "bin", "long-to-linked-pe")):
        try:
            if open(host_stamp).read().strip() == host_digest:
                return HOST_OUT
        except OSError:
            if all(os.path.getmtime(p) <= os.path.getmtime(HOST_OUT) for p in HOST_SOURCES + [OUT]):
                return HOST_OUT
    os.makedirs(os.path.dirname(HOST_OUT), exist_ok=True)
    cxx = os.environ.get("CXX", "g++")
    cmd = [cxx, "-O2", "-std=c++17", "-pthread", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(HERE, "host"), HOST_SOURCES[0],
           "-L" + os.path.dirname(OUT), "-larks_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lz", "-ldl",
           "-Wl,-rpath,$ORIGIN/../lib", "-Wl,-rpath,/opt/rocm/lib", "-o", HOST_OUT]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    # the arks-long feeder (no GPU code)
    cmd2 = [cxx, "-O2", "-std=c++17", "-pthread", "-I" + os.path.join(HERE, "host"),
            os.path.join(HERE, "host", "long_to_linked_pe.cpp"), "-lz", "-ldl",
            "-o", os.path.join(HERE, "bin", "long-to-linked-pe")]
    if verbose:
        print(" ".join(cmd2), file=sys.stderr)
    subprocess.check_call(cmd2)
    with open(host_stamp, "w") as f:
        f.write(host_digest + "\n")
    return HOST_OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_host(force="--force" in sys.argv, verbose=True))
