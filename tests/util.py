"""helpers shared by the tests"""
import hashlib

import numpy as np


def index_digest(keys, vals):
    """order-independent digest of an exported index (same recipe as tests/golden/make_golden.py)"""
    keys = np.ascontiguousarray(keys)
    order = np.lexsort(keys.T[::-1]) if len(keys) else np.zeros(0, dtype=np.int64)
    h = hashlib.sha256()
    h.update(keys[order].tobytes())
    h.update(np.asarray(vals)[order].astype("<i4").tobytes())
    return h.hexdigest()


def oracle_pairs(oracle_mod, ox, reads, pair_ok, barcode, j):
    """chromiumRead's pair flow through the oracle: (conreci, pair, stats, triples)"""
    data = "".join(reads).encode()
    lens = np.array([len(r) for r in reads], dtype=np.uint32)
    offsets = np.zeros(len(reads), dtype=np.uint64)
    if len(reads):
        offsets[1:] = np.cumsum(lens[:-1])
    conreci, pair, st = ox.map_pairs(data + b"\0", offsets, lens, j,
                                     pair_ok=np.asarray(pair_ok, dtype=np.uint8))
    imap = {}
    for p, c in enumerate(pair):
        if c and pair_ok[p]:
            imap[(int(barcode[p]), int(c))] = imap.get((int(barcode[p]), int(c)), 0) + 1
    triples = sorted([b, c, n] for (b, c), n in imap.items())
    return conreci, pair, st, triples
