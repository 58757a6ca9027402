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


@pytest.mark.parametrize("how", ["auto", "blocks", "round_robin"])
@pytest.mark.parametrize("n_yaw,n_pitch,world", [(12, 3, 8), (12, 3, 2), (12, 3, 1), (4, 1, 3), (1, 1, 4), (5, 7, 6), (360, 1, 8)])
def test_rank_view_sets_cover_every_view_exactly_once(n_yaw, n_pitch, world, how):
    """The masked job of each rank (what bench.py --scaling strong and process_views_sharded build): the ranks' masks,
    mapped back to the image's view indices, are a partition of the yaw x pitch grid, sized within one of each other."""
    import importlib

    import numpy as np

    sys.path.insert(0, ROOT)
    drv = importlib.import_module("360-to-planer-images_amd._driver")
    seen = np.zeros((n_yaw, n_pitch), np.int32)
    sizes = []
    for rank in range(world):
        yaw_idx, pitch_idx, mask, mine = drv.rank_view_set(n_yaw, n_pitch, world, rank, how)
        assert mask.shape == (len(yaw_idx), len(pitch_idx)) and mask.dtype == np.uint8
        assert yaw_idx == sorted(set(yaw_idx)) and pitch_idx == sorted(set(pitch_idx))
        assert int(mask.sum()) == len(mine) == len(set(mine))
        if mine:  # every yaw and pitch the job is created with draws at least one view
            assert mask.any(axis=1).all() and mask.any(axis=0).all()
        for a, y in enumerate(yaw_idx):
            for b, p in enumerate(pitch_idx):
                assert bool(mask[a, b]) == ((y, p) in mine)
                seen[y, p] += int(mask[a, b])
        assert mine == sorted(mine, key=lambda v: (v[1], v[0]))  # pitch-major: the order the views are downloaded in
        sizes.append(len(mine))
    assert (seen == 1).all()
    assert sum(sizes) == n_yaw * n_pitch
    if how in ("blocks", "round_robin"):
        assert max(sizes) - min(sizes) <= 1  # (cut by cost, a run across a pitch boundary holds fewer views)


def test_rank_view_set_of_config_2_on_8_gpus():
    import importlib

    sys.path.insert(0, ROOT)
    drv = importlib.import_module("360-to-planer-images_amd._driver")
    yaw_idx, pitch_idx, mask, mine = drv.rank_view_set(12, 3, 8, 0)  # views 0..4 of the pitch-major list
    assert mine == [(0, 0), (1, 0), (2, 0), (3, 0), (4, 0)]
    assert yaw_idx == [0, 1, 2, 3, 4] and pitch_idx == [0] and mask.all()    # five yaws of one pitch view: no mask needed
    yaw_idx, pitch_idx, mask, mine = drv.rank_view_set(12, 3, 8, 2, "blocks")  # cut by count: the run that crosses from pitch 0 to pitch 1
    assert mine == [(10, 0), (11, 0), (0, 1), (1, 1), (2, 1)]
    assert yaw_idx == [0, 1, 2, 10, 11] and pitch_idx == [0, 1]
    assert mask.tolist() == [[0, 1], [0, 1], [0, 1], [1, 0], [1, 0]]         # 5 of a 5 x 2 grid
    yaw_idx, pitch_idx, mask, mine = drv.rank_view_set(12, 3, 8, 2, pitch_deg=[60, 90, 120])  # cut by cost (the default): two set-ups, 3 views
    assert mine == [(10, 0), (11, 0), (0, 1)] and yaw_idx == [0, 10, 11] and pitch_idx == [0, 1]
    assert mask.tolist() == [[0, 1], [1, 0], [1, 0]]
    yaw_idx, pitch_idx, mask, mine = drv.rank_view_set(12, 3, 8, 0, "round_robin")  # (the dealing before: 5 of a 3 x 3 grid)
    assert mine == [(0, 0), (8, 0), (4, 1), (0, 2), (8, 2)] and mask.tolist() == [[1, 0, 1], [0, 1, 0], [1, 0, 1]]
    assert drv.rank_view_set(2, 1, 4, 3)[3] == []                            # more ranks than views


@pytest.mark.parametrize("oh,world", [(1080, 8), (1080, 3), (800, 8), (270, 4), (40, 8), (16, 2), (4096, 8), (17, 1), (1, 4)])
def test_row_bands_cover_every_row_exactly_once(oh, world):
    """_driver.shard_rows (one image's rows shared out to the GPUs: p2p_job_set_rows): contiguous bands of whole tile
    rows that cover [0, oh) once, in rank order; surplus ranks get empty bands."""
    import importlib
    import numpy as np
    d = importlib.import_module("360-to-planer-images_amd._driver")
    for pitches, fov in (([60, 90, 120], 90), ([30, 60, 90, 120, 150], 90), (None, 60), ([5], 150)):
        bands = d.shard_rows(oh, world, pitches, fov, 2 * oh)
        assert len(bands) == world
        assert bands[0][0] == 0 and max(b for _, b in bands) == oh
        for (a0, a1), (b0, b1) in zip(bands[:-1], bands[1:]):
            assert a1 == b0 and a0 <= a1
        for a, b in bands:
            assert a == b or (a % d.TILE_ROWS == 0 and (b % d.TILE_ROWS == 0 or b == oh))
        covered = np.zeros(oh, int)
        for a, b in bands:
            covered[a:b] += 1
        assert (covered == 1).all()


def test_row_bands_of_config_2_on_8_gpus_are_balanced():
    import importlib
    d = importlib.import_module("360-to-planer-images_amd._driver")
    bands = d.shard_rows(1080, 8, [60, 90, 120], 90, 1920)
    sizes = [b - a for a, b in bands]
    assert max(sizes) <= 144 and min(sizes) >= 112, sizes  # 68 tile rows on 8 ranks: 7 to 9 each
