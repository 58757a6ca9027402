"""The N > 1 path of bench.py on CPU: two gloo ranks run the sharding, the barrier-bracketed timed loop and
the max-over-ranks reduction (the view batch shards with no data-path collective, so that is all there is)."""
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import time

    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import bench

    d = bench.Dist(backend="gloo")
    mine = bench.shard_round_robin(64, world, rank)  # config 3: 64 panoramas dealt to the ranks
    steps = []

    def step():
        steps.append(1)
        time.sleep(0.01 * (rank + 1))  # rank 1 is the slow one

    elapsed = bench.run_timed(step, lambda: None, d, steps=5, warmup=2)
    q.put((rank, mine, len(steps), elapsed, d.max_over_ranks(rank * 10.0)))
    d.close()


def test_two_rank_gloo_sharding_and_timing():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, m0, n0, e0, x0), (r1, m1, n1, e1, x1) = res
    assert sorted(m0 + m1) == list(range(64)) and not set(m0) & set(m1)   # every panorama exactly once
    assert m0 == list(range(0, 64, 2)) and m1 == list(range(1, 64, 2))   # round-robin
    assert n0 == n1 == 7                                                  # W + K steps on every rank
    assert e0 == e1 and e0 >= 5 * 0.02 * 0.9                              # MAX over ranks = the slow rank
    assert x0 == x1 == 10.0


def test_round_robin_edge_cases():
    sys.path.insert(0, ROOT)
    import bench

    assert bench.shard_round_robin(3, 8, 5) == []           # fewer items than ranks
    assert bench.shard_round_robin(36, 8, 0) == [0, 8, 16, 24, 32]  # 36 views on 8 GPUs: 5 vs 4 (cap 7.2x)
    assert sum(len(bench.shard_round_robin(36, 8, r)) for r in range(8)) == 36
    w = bench.WORKLOADS["cfg2"]
    assert bench.algorithmic_bytes(w, 1) == 3 * 8192 * 4096 + 3 * 1920 * 1080 * 36 == 324612096
