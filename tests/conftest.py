import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device")


# Order of the GPU suite: the committed golden fixtures and the demo-log counters first (configs[0] on the HIP path,
# the vote rules), then the oracle comparisons of the other kernels, the long full-size runs after them, the
# many-process stress test last.  `pytest -x` (the driver's flags) then stops at the most specific failure.
_ORDER = ["test_gpu_parity", "test_gpu_imap", "test_gpu_repeats", "test_gpu_sharded", "test_gpu_seed_shards",
          "test_gpu_exchange", "test_gpu_fuzz", "test_gpu_cli", "test_gpu_bench", "test_gpu_full_size", "test_zz_gpu_stress"]


def pytest_collection_modifyitems(config, items):
    def rank(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return _ORDER.index(name) if name in _ORDER else -1          # CPU tests: in front, in their own order
    items.sort(key=rank)                                             # (stable: the order inside a file stays)


def read_fasta(path):
    seqs, name, cur = [], None, []
    with open(path) as fh:
        for line in fh:
            line = line.rstrip("\n")
            if line.startswith(">"):
                if name is not None:
                    seqs.append((name, "".join(cur)))
                name, cur = line[1:].split()[0], []
            else:
                cur.append(line)
    if name is not None:
        seqs.append((name, "".join(cur)))
    return seqs


@pytest.fixture(scope="session")
def oracle():
    """the CPU oracle (test infrastructure), built on demand with gcc"""
    from oracle import pyoracle
    pyoracle.build_oracle()
    return pyoracle


@pytest.fixture(scope="session")
def demo_contigs():
    return read_fasta(os.path.join(GOLDEN, "arks_test-demo.test_scaffolds.fa"))


@pytest.fixture(scope="session")
def golden_keys():
    return json.load(open(os.path.join(GOLDEN, "keys.json")))


@pytest.fixture(scope="session")
def golden_demo_index():
    return json.load(open(os.path.join(GOLDEN, "demo_index.json")))


@pytest.fixture(scope="session")
def golden_mini():
    return json.load(open(os.path.join(GOLDEN, "mini.json")))


@pytest.fixture(scope="session")
def arks():
    """the product library; GPU tests fail (not skip) when it is missing"""
    import arcs_amd
    from arcs_amd import build as b
    if b.needs_build():          # hipcc cross-compiles gfx950 without a GPU; built in-tree
        b.build()
    arcs_amd.lib()
    return arcs_amd


@pytest.fixture(scope="session")
def gpu(arks):
    import torch
    assert torch.cuda.is_available(), "gpu-marked test on a machine without a GPU"
    assert arks.device_count() >= 1, "no gfx950 device visible to libarks_hip"
    return 0


@pytest.fixture
def medium_blocks(arks):
    """arks_debug_set_medium_blocks for the length of one test (the cap is process-wide)"""
    yield arks.api.set_medium_blocks
    arks.api.set_medium_blocks(0)


@pytest.fixture(params=["seeds", "minimizer"])
def index_layout(request, monkeypatch):
    """both layouts of the locality index: the seed index (every m-mer position in the table, fixed seeds
    on the read side; arks_index_kind 2) and the minimizer index (kind 1); results must be identical"""
    from arcs_amd import api
    monkeypatch.setitem(api.BUILD_DEFAULTS, "index_kind", request.param)      # arks_build_options.index_kind
    return request.param
