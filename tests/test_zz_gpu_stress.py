"""The stress test of the GPU suite, in a file that sorts LAST (and tests/conftest.py orders it last whatever the
names): under `pytest -x` a failure here can no longer hide a parity result (round 5: it sat in front of
test_gpu_parity.py and took 104 tests with it).  Children run under `python -X faulthandler`: a host-side fault leaves
the python stack of every thread in the assertion message."""
import pytest

pytestmark = [pytest.mark.gpu]


def test_many_processes_share_the_device(arks, gpu, oracle):
    """Stress: 12 processes on the one device at once, each building a small index and mapping against it case after
    case (tests/fuzz_open_ended.py: all index layouts, the contig-sharded and the seed-sharded paths) for 45 s -- what
    `arcs --ranks` and a shared node do to the library.  Every process must end with `fuzz ok` (a GPU memory fault
    kills the process: round 2 saw two such deaths in a 40-minute run of 24 processes, before the kernels lost their
    scratch use)."""
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FUZZ_SHARDS="1", FUZZ_SEED_SHARDS="1")
    procs = []
    with tempfile.TemporaryDirectory() as tmp:
        for p in range(12):
            e = dict(env, FUZZ_TRACE=os.path.join(tmp, f"seed{p}"))
            procs.append(subprocess.Popen([sys.executable, "-X", "faulthandler", os.path.join(root, "tests", "fuzz_open_ended.py"), "45",
                                           str(700_000_000 + 1_000_000 * p)], cwd=root, env=e,
                                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        outs = [p.communicate(timeout=600) for p in procs]
        last = [open(os.path.join(tmp, f"seed{p}")).read() if os.path.exists(os.path.join(tmp, f"seed{p}")) else "?"
                for p in range(12)]
    for p, (proc, (out, err)) in enumerate(zip(procs, outs)):
        assert proc.returncode == 0 and "fuzz ok" in out, (p, proc.returncode, "phase (a case number, 'done', 'exit'): " + last[p],
                                                           "stdout: " + out[-300:], err[-6000:])
    cases = sum(int(o.split("fuzz ok:")[1].split("cases")[0]) for o, _ in outs)
    assert cases >= 200, cases
