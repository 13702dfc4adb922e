"""bench.py's multi-rank control flow on ONE GPU: `python bench.py --gpus N` starts its N ranks itself
(torch.distributed.run), the one read set is dealt to the ranks in blocks (strong scaling), and the summed
counters and stored pairs of the N-rank run equal the one-rank run's.  ARKS_BENCH_BACKEND=gloo lets the ranks
share the box's single device; the driver's runs use nccl (RCCL), one rank per GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--steps", "2", "--warmup", "1", "--pairs", "400000", "--chunk", "150000", "--draft-mbp", "5",
        "--no-cpu-baseline", "--no-extras"]


def _bench(extra, env_extra=None):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + ARGS + extra, env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout          # ONE JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_gpus_flag_starts_the_ranks(arks, gpu):
    one = _bench([])
    assert one["n_gpus"] == 1 and one["scaling"] == "strong"
    assert one["config"]["pairs_job"] == 400000
    r = one["roofline"]
    assert r["frac"] is None or 0.0 < r["frac"] <= 1.0       # a fraction (None: no counter summary for this workload)
    assert r["alg_frac"] > 0
    for n in (2, 3):
        many = _bench(["--gpus", str(n)], {"ARKS_BENCH_BACKEND": "gloo"})
        assert many["n_gpus"] == n and many["scaling"] == "strong"
        assert many["config"]["pairs_job"] == 400000
        assert many["counters"] == one["counters"]            # the same pairs, whoever maps them
        assert many["stored_pairs"] == one["stored_pairs"]
        assert many["config"]["launches_per_step"] >= 1
    weak = _bench(["--gpus", "2", "--weak"], {"ARKS_BENCH_BACKEND": "gloo"})
    assert weak["n_gpus"] == 2 and weak["scaling"] == "weak"
    assert weak["config"]["pairs_job"] == 800000
    assert weak["counters"]["windows"] > one["counters"]["windows"]


@pytest.mark.gpu
def test_bench_sharded_index_over_the_exchange(arks, gpu):
    """`bench.py --sharded-index` with one process: the seed table in 3 local shards, the read set dealt to them,
    every batch through arks_exchange -- the summed counters equal the replica run's on the same read set"""
    one = _bench([])
    sh = _bench(["--sharded-index", "--shards", "3"])
    assert sh["n_gpus"] == 1 and sh["config"]["shards"] == 3 and sh["scaling"] == "strong"
    assert sh["counters"] == one["counters"]
    ex = sh["config"]["last_batch_of_rank0"]
    assert ex["seeds"] > 0 and 0 < ex["sent"] < ex["seeds"] and ex["received"] > 0
    assert max(sh["config"]["shard_bytes"]) > 0


@pytest.mark.gpu
def test_bench_eight_ranks_dry_run(arks, gpu):
    """What the driver will run the day an 8-GPU node appears -- `python bench.py --gpus 8` -- as a dry run on the one
    GPU (ARKS_BENCH_BACKEND=gloo: eight processes share it): the same control flow (rendezvous, blocks dealt to eight
    ranks, max-over-ranks timing, summed counters, one JSON line), counters equal to one rank's, and a wall-time
    budget: eight processes generate the draft and build their index replicas side by side on the box's CPU quota,
    which must stay far inside the driver's limit."""
    import time
    one = _bench([])
    t0 = time.time()
    many = _bench(["--gpus", "8"], {"ARKS_BENCH_BACKEND": "gloo"})
    wall = time.time() - t0
    assert many["n_gpus"] == 8 and many["scaling"] == "strong"
    assert many["config"]["pairs_job"] == 400000
    assert many["counters"] == one["counters"] and many["stored_pairs"] == one["stored_pairs"]
    assert "x8" in many["config"]["parallelism"]
    assert wall < 600, wall
