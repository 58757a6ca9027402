"""GPU parity of the work lists (csrc/p2p_host_plan.cpp: xcd_main_lists, xcd_lists): which XCD draws which tile, and when,
must not change a byte.  Every combination of main-kernel order (grid / list / list for several panoramas), turn length
per chunk of pairs, table-prefetch workgroups and gather-tile order is checked against the CPU restatement of the
reference's two cv2.remap stages (P:181-221) on a job that has LDS-scheme tiles, gather tiles (a pole in view),
several chunks of pairs and a flickering yaw -- in both tile shapes the library is built with."""
import itertools

import numpy as np
import pytest

from _util import oracle_maps, oracle_views, poison_views

pytestmark = pytest.mark.gpu


def _run(gpu, panos, yaws, pitches, ow, oh, fov, maps):
    ph, pw = panos[0].shape[:2]
    ctx = gpu.Context(0)
    try:
        job = gpu.Job(ctx, pw, ph, len(panos), yaws, pitches, fov, ow, oh)
        if maps is not None:
            job.set_maps(*maps)
        for i, p in enumerate(panos):
            job.set_pano(i, p)
        poison_views(job)  # (a tile that no list holds must show)
        job.run()
        out = [job.get_views(i) for i in range(len(panos))]
        job.close()
        return out
    finally:
        ctx.close()


@pytest.mark.parametrize("tile_shape", ["64", "128"])
@pytest.mark.parametrize("n_panos", [1, 2])
def test_every_work_list_order_draws_the_oracles_bytes(gpu, synth, monkeypatch, n_panos, tile_shape, p2p_env):
    pw, ph, ow, oh, fov = 2048, 1024, 333, 210, 90
    yaws = [0, 14.0625, 33, 90, 123.4, 180, 200, 270, 301, 359]   # whole-column, fractional and (14.0625 on 2048: none) plain ones
    pitches = [8, 60, 90, 150]                                    # pitch 8: a pole in view -> gather tiles
    # TWO sets of panoramas, used in turn: a job's output block comes out of the device memory pool, where the run before
    # left its views -- with one set a tile that no list holds would still show the right pixels
    sets = [[synth.synth_pano(pw, ph, 4200 + 10 * k + i, "N") for i in range(n_panos)] for k in range(2)]
    maps = oracle_maps(yaws, pitches, ow, oh, pw, ph, fov)
    wants = [[oracle_views(p, yaws, pitches, ow, oh, fov) for p in panos] for panos in sets]
    p2p_env("P2P_PLAN_CACHE", "0")
    p2p_env("P2P_TILE_SHAPE", tile_shape)  # both tile shapes of the library (csrc/p2p_device.h)
    p2p_env("P2P_PAIRS_PER_BLOCK", "3")   # several chunks of pairs per tile
    combos = list(itertools.product(("0", "1", "2"), ("1", "5", "192"), ("0", "1"), ("0", "1")))
    for k, (main_order, group, prefetch, gather_order) in enumerate(combos):
        p2p_env("P2P_MAIN_ORDER", main_order)
        p2p_env("P2P_MAIN_GROUP", group)
        p2p_env("P2P_PREFETCH_LEAD", prefetch)
        p2p_env("P2P_GATHER_ORDER", gather_order)
        got = _run(gpu, sets[k & 1], yaws, pitches, ow, oh, fov, maps)
        for i in range(n_panos):
            bad = np.argwhere(got[i] != wants[k & 1][i])
            assert bad.size == 0, (tile_shape, main_order, group, prefetch, gather_order, i, len(bad), bad[:3])


@pytest.mark.parametrize("n_panos", [1, 2])
def test_workgroups_that_draw_several_chunks_of_pairs_draw_the_oracles_bytes(gpu, synth, monkeypatch, n_panos, p2p_env):
    """P2P_MAIN_SPAN: one workgroup of the 128-wide main kernel draws 2, 3 ... all chunks of pairs of its tile in turn
    (what config 4 ships with: its plan tables are read once per tile instead of once per chunk) -- in grid and list
    order, with and without the table-prefetch workgroups, with a view mask that empties whole chunks."""
    pw, ph, ow, oh, fov = 2048, 1024, 333, 210, 90
    yaws = [0, 14.0625, 33, 90, 123.4, 180, 200, 270, 301, 359]
    pitches = [8, 60, 90, 150]
    panos = [synth.synth_pano(pw, ph, 4300 + i, "N") for i in range(n_panos)]
    maps = oracle_maps(yaws, pitches, ow, oh, pw, ph, fov)
    want = [oracle_views(p, yaws, pitches, ow, oh, fov) for p in panos]
    p2p_env("P2P_PLAN_CACHE", "0")
    p2p_env("P2P_TILE_SHAPE", "128")
    p2p_env("P2P_PAIRS_PER_BLOCK", "3")   # 4 chunks of pairs per panorama's 10 yaws (7 for two panoramas)
    for span, main_order, prefetch in itertools.product(("2", "3", "4", "64"), ("0", "1", "2"), ("0", "1")):
        p2p_env("P2P_MAIN_SPAN", span)
        p2p_env("P2P_MAIN_ORDER", main_order)
        p2p_env("P2P_PREFETCH_LEAD", prefetch)
        got = _run(gpu, panos, yaws, pitches, ow, oh, fov, maps)
        for i in range(n_panos):
            bad = np.argwhere(got[i] != want[i])
            assert bad.size == 0, (span, main_order, prefetch, i, len(bad), bad[:3])
    # a view mask that leaves the second chunk (yaws 3..5) without a wanted view at pitch 60, and the first at pitch 90
    p2p_env("P2P_MAIN_SPAN", "4")
    p2p_env("P2P_MAIN_ORDER", "1")
    mask = np.ones((len(yaws), len(pitches)), np.uint8)
    mask[3:6, 1] = 0
    mask[0:3, 2] = 0
    ctx = gpu.Context(0)
    try:
        job = gpu.Job(ctx, pw, ph, 1, yaws, pitches, fov, ow, oh)
        job.set_maps(*maps)
        job.set_view_mask(mask)
        job.set_pano(0, panos[0])
        assert job.info()["chunks_per_workgroup"] == 4 and job.info()["pair_chunks"] == 4
        poison_views(job)
        job.run()
        got = job.get_views(0)
        job.close()
    finally:
        ctx.close()
    for y, p in itertools.product(range(len(yaws)), range(len(pitches))):
        if mask[y, p]:
            assert np.array_equal(got[y, p], want[0][y, p]), (y, p)


@pytest.mark.parametrize("tile_shape", ["64", "128"])
def test_tail_entries_drawn_by_several_workgroups_draw_the_oracles_bytes(gpu, synth, monkeypatch, tile_shape, p2p_env):
    """P2P_MAIN_TAIL / P2P_MAIN_TAIL_PARTS: the last entries of every XCD's main list are drawn by 2..4 workgroups, a
    part of the job's pairs each (what config 2 ships with: 44 entries, two workgroups) -- no tail, one entry, more
    entries than the lists hold; odd pair counts; a view mask that empties one part."""
    pw, ph, ow, oh, fov = 2048, 1024, 333, 210, 90
    pitches = [8, 60, 90, 150]
    for yaws in ([0, 14.0625, 33, 90, 123.4, 180, 200, 270, 301, 359], [0, 33, 90, 200, 301]):
        pano = synth.synth_pano(pw, ph, 4400 + len(yaws), "N")
        maps = oracle_maps(yaws, pitches, ow, oh, pw, ph, fov)
        want = oracle_views(pano, yaws, pitches, ow, oh, fov)
        p2p_env("P2P_PLAN_CACHE", "0")
        p2p_env("P2P_TILE_SHAPE", tile_shape)
        p2p_env("P2P_MAIN_ORDER", "1")
        for tail, parts in itertools.product(("0", "1", "7", "100000"), ("2", "3", "4")):
            p2p_env("P2P_MAIN_TAIL", tail)
            p2p_env("P2P_MAIN_TAIL_PARTS", parts)
            got = _run(gpu, [pano], yaws, pitches, ow, oh, fov, maps)[0]
            bad = np.argwhere(got != want)
            assert bad.size == 0, (tile_shape, len(yaws), tail, parts, len(bad), bad[:3])
    # the mask leaves the second half of the yaws without a wanted view at pitch 60
    p2p_env("P2P_MAIN_TAIL", "100000")
    p2p_env("P2P_MAIN_TAIL_PARTS", "2")
    mask = np.ones((len(yaws), len(pitches)), np.uint8)
    mask[3:, 1] = 0
    ctx = gpu.Context(0)
    try:
        job = gpu.Job(ctx, pw, ph, 1, yaws, pitches, fov, ow, oh)
        job.set_maps(*maps)
        job.set_view_mask(mask)
        job.set_pano(0, pano)
        poison_views(job)
        job.run()
        got = job.get_views(0)
        job.close()
    finally:
        ctx.close()
    for y, p in itertools.product(range(len(yaws)), range(len(pitches))):
        if mask[y, p]:
            assert np.array_equal(got[y, p], want[y, p]), (y, p)


def test_device_maps_job_is_the_same_in_every_order_and_tile_shape(gpu, synth, monkeypatch, p2p_env):
    # the default path (maps evaluated on the device): list order, grid order and both tile shapes agree byte for byte
    pw, ph, ow, oh, fov = 4096, 2048, 640, 360, 90
    yaws, pitches = list(range(0, 360, 20)), [45, 90, 135]
    pano = synth.synth_pano(pw, ph, 4300, "N")
    p2p_env("P2P_PLAN_CACHE", "0")
    outs = []
    for tile_shape in ("64", "128"):
        for main_order in ("0", "1"):
            p2p_env("P2P_TILE_SHAPE", tile_shape)
            p2p_env("P2P_MAIN_ORDER", main_order)
            outs.append(_run(gpu, [pano], yaws, pitches, ow, oh, fov, None)[0])
    for o in outs[1:]:
        assert np.array_equal(outs[0], o)


@pytest.mark.parametrize("blocky_from", ["0", "12", "1000000"])
def test_gather_tiles_in_rows_and_in_blocks_draw_the_same_bytes(gpu, synth, monkeypatch, blocky_from, p2p_env):
    # the gather kernel's lane layout (rows of 64 pixels / blocks of 16 x 4, csrc/p2p_views.hip: draw_gather) is chosen
    # per tile by the plan; forcing every tile into blocks (0), none (a huge threshold) or the default must not change
    # a byte: polar views (every tile gathers around the pole), a strongly minifying view set (every tile gathers),
    # view sizes that are not multiples of the tile
    p2p_env("P2P_PLAN_CACHE", "0")
    p2p_env("P2P_GATHER_BLOCKY_FROM", blocky_from)
    for (pw, ph, ow, oh, fov, yaws, pitches) in (
            (2048, 1024, 301, 177, 90, [0, 33.3, 180, 270], [3, 90, 176]),
            (4096, 2048, 203, 150, 110, [0, 90, 200], [30, 60, 150])):
        pano = synth.synth_pano(pw, ph, 4400 + ow, "N")
        maps = oracle_maps(yaws, pitches, ow, oh, pw, ph, fov)
        want = oracle_views(pano, yaws, pitches, ow, oh, fov)
        for tile_shape in ("64", "128"):
            p2p_env("P2P_TILE_SHAPE", tile_shape)
            got = _run(gpu, [pano], yaws, pitches, ow, oh, fov, maps)[0]
            bad = np.argwhere(got != want)
            assert bad.size == 0, (blocky_from, tile_shape, pw, ow, len(bad), bad[:3])


@pytest.mark.parametrize("pitches", [[60, 90, 120], [8, 60, 90]], ids=["no gather tile", "a pole in view"])
def test_first_and_second_launch_draw_the_same_bytes(gpu, synth, p2p_env, pitches):
    """A job's first launch sends the main kernel out behind the plan pass without waiting for anything: in grid order,
    with the per-XCD lists (and the job's pair contexts) made on the device behind the gather count's copy while the host
    wakes up (P2P_DEFER_LISTS=1, the default) or when a second launch asks (2), or in list order with both made in between
    (0), or after the gather count has come back (P2P_EARLY_MAIN=0).  The first, the second and the third launch -- where
    the gather tiles ride in the main kernel's launch (P2P_MERGE_GATHER) -- draw the oracle's bytes into a poisoned block,
    with every knob every way."""
    pw, ph, ow, oh, fov = 2048, 1024, 640, 360, 90
    yaws = [0, 33, 90, 200, 301]
    pano = synth.synth_pano(pw, ph, 4400, "N")
    maps = oracle_maps(yaws, pitches, ow, oh, pw, ph, fov)
    want = oracle_views(pano, yaws, pitches, ow, oh, fov)
    p2p_env("P2P_PLAN_CACHE", "0")
    for early, defer, merge, table in itertools.product(("1", "0"), ("1", "0", "2"), ("1", "0"), ("1", "0")):
        p2p_env("P2P_EARLY_MAIN", early)
        p2p_env("P2P_DEFER_LISTS", defer)
        p2p_env("P2P_MERGE_GATHER", merge)  # list order: the gather tiles as the first workgroups of the main kernel's launch
        p2p_env("P2P_PAIR_CTX_TABLE", table)  # the tiles' pair contexts from the job's table (from the second launch on) or worked out per workgroup
        ctx = gpu.Context(0)
        try:
            job = gpu.Job(ctx, pw, ph, 1, yaws, pitches, fov, ow, oh)
            job.set_maps(*maps)
            job.set_pano(0, pano)
            for launch in range(3):
                poison_views(job, 0xA5 - launch)  # (a tile that a launch does not draw must show)
                job.run()
                got = job.get_views(0)
                bad = np.argwhere(got != want)
                assert bad.size == 0, (early, defer, merge, table, launch, len(bad), bad[:3])
            job.close()
        finally:
            ctx.close()
