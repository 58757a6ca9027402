"""GPU, BASELINE.json's full sizes, through size-independent properties (the CPU oracle would take minutes
per case here, so only one yaw of config 2 is compared pixel by pixel):
  * a whole-column yaw shift equals rolling the panorama (exact);
  * a constant panorama gives constant views (the weights sum to 1 in both stages);
  * caller-map mode fed the fused kernel's own coordinates reproduces the fused output (exact);
  * one yaw of config 2 against the oracle: fused within +-1, caller-map mode bit-exact."""
import numpy as np
import pytest

from _util import coords_to_maps, diff_stats, oracle_maps, oracle_views
from oracle import maps

pytestmark = pytest.mark.gpu

CFG2 = dict(pw=8192, ph=4096, ow=1920, oh=1080, fov=90, pitches=[60, 90, 120])


@pytest.fixture(scope="module")
def pano8k(synth):
    return synth.synth_pano(CFG2["pw"], CFG2["ph"], 1000, "S")


def test_cfg2_one_yaw_vs_oracle(gpu, pkg, pano8k):
    c = CFG2
    want = oracle_views(pano8k, [30], c["pitches"], c["ow"], c["oh"], c["fov"])
    fused = pkg.process_views(pano8k, [30], c["pitches"], c["ow"], c["oh"], c["fov"])
    mx, gt1, anyd = diff_stats(fused, want)
    print("cfg2 yaw 30 fused vs oracle: max %d, >1: %.3g, any: %.3g" % (mx, gt1, anyd))
    assert mx <= 1
    rows, U, V = oracle_maps([30], c["pitches"], c["ow"], c["oh"], c["pw"], c["ph"], c["fov"])
    assert np.array_equal(gpu.remap_views_maps(pano8k, rows, U, V), want)


def test_cfg2_whole_column_shift_equals_roll(gpu, pkg, pano8k):
    c = CFG2
    a = pkg.process_views(pano8k, [45, 180], c["pitches"], c["ow"], c["oh"], c["fov"])
    for k, yaw in enumerate((45, 180)):
        shift = yaw * c["pw"] // 360
        b = pkg.process_views(np.roll(pano8k, -shift, axis=1), [0], c["pitches"], c["ow"], c["oh"], c["fov"])
        assert np.array_equal(a[k], b[0])


def test_cfg2_all_36_views_self_consistent(gpu, pkg, pano8k):
    c = CFG2
    yaws = list(range(0, 360, 30))
    ctx = gpu.Context(0)
    job = gpu.Job(ctx, c["pw"], c["ph"], 1, yaws, c["pitches"], c["fov"], c["ow"], c["oh"], flags=gpu.FLAG_KEEP_COORDS)
    job.set_pano(0, pano8k)
    job.run()
    fused, coords = job.get_views(0), job.get_coords()
    job.close()
    U, V = zip(*(coords_to_maps(coords[p]) for p in range(3)))
    rows = np.stack([maps.yaw_column_table(c["pw"], y) for y in yaws])
    again = gpu.remap_views_maps(pano8k, rows, np.stack(U), np.stack(V))
    assert np.array_equal(fused, again)  # 74.6 Mpix, every byte
    ctx.close()


def test_constant_panorama_constant_views_cfg4_size(gpu, pkg):
    # config 4 geometry (16384x8192 -> 4096x4096, FOV 60) on a few views incl. the pole-containing pitch 30
    pano = np.empty((8192, 16384, 3), np.uint8)
    pano[:] = (7, 130, 251)
    v = pkg.process_views(pano, [5, 200], [30, 90], 4096, 4096, 60)
    assert (v == np.array([7, 130, 251], np.uint8)).all()
