"""The identical-results route as a product mode: the tool run as a SCRIPT with --exact on NOISE panoramas writes files
whose pixels are the oracle's, byte for byte -- BASELINE config 1, a 3-view sample of config 2, the
reference CLI's default view set (P:412-437), a folder through the two-slot pipeline, one image shared out to several
contexts; and process_yaw_and_pitchs(..., exact=True) from the reference's own thread fan-out (P:252-265).
Noise is the hard case: with device-evaluated maps 0.001-0.017 % of the 1/32-px quantisations fall the other way and a noise
pixel then differs by up to 8 levels (tests/test_gpu_views_fused.py); here nothing may differ.
The maps come from the package's own NumPy evaluation (_exact_maps.py, pinned to the reference-made goldens by
tests/test_exact_maps.py); every pixel is drawn on the GPU."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests._util import oracle_views, oracle_views_threaded

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "360-to-planer-images_amd", "panorama_to_plane_pitch.py")


def _run(args, cwd):
    return subprocess.run([sys.executable, TOOL] + args, cwd=cwd, capture_output=True, text=True, timeout=900)


def _save(path, arr):
    from PIL import Image
    Image.fromarray(arr).save(path, compress_level=1)


def _load(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"))


def _check_folder(out_dir, stem, pano, yaws, pitches, ow, oh, fov=90, threaded=False):
    want = (oracle_views_threaded if threaded else oracle_views)(pano, yaws, pitches, ow, oh, fov)
    names = sorted(os.listdir(out_dir))
    assert len([n for n in names if n.startswith(stem + "_")]) == len(yaws) * len(pitches), names[:4]
    for yi, y in enumerate(yaws):
        for pi, p in enumerate(pitches):
            got = _load(os.path.join(out_dir, f"{stem}_{ow}x{oh}_yaw_{y}_pitch_{p}.png"))
            assert np.array_equal(got, want[yi, pi]), (stem, y, p, int(np.abs(got.astype(int) - want[yi, pi]).max()))


def test_cli_exact_config1_noise(tmp_path, synth):
    """BASELINE config 1: 2048 x 1024 -> 512 x 512, FOV 90, yaw 0, pitch 90."""
    pano = synth.synth_pano(2048, 1024, 1000, "N")
    _save(tmp_path / "c1.png", pano)
    r = _run(["--input_path", str(tmp_path / "c1.png"), "--output_path", str(tmp_path / "o"), "--exact", "--yaw_angles", "0",
              "--pitch_angles", "90", "--output_width", "512", "--output_height", "512"], tmp_path)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    _check_folder(tmp_path / "o", "c1", pano, [0], [90], 512, 512)


def test_cli_exact_config2_sample_noise(tmp_path, synth):
    """Three views of BASELINE config 2 (8192 x 4096 -> 1920 x 1080): a fractional yaw shift, all three pitches."""
    pano = synth.synth_pano(8192, 4096, 1000, "N")
    _save(tmp_path / "c2.png", pano)
    r = _run(["--input_path", str(tmp_path / "c2.png"), "--output_path", str(tmp_path / "o"), "--exact", "--yaw_angles", "30",
              "--pitch_angles", "60", "90", "120", "--output_width", "1920", "--output_height", "1080"], tmp_path)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    _check_folder(tmp_path / "o", "c2", pano, [30], [60, 90, 120], 1920, 1080)


def test_cli_exact_reference_defaults_noise(tmp_path, synth):
    """The reference CLI's defaults (P:412-437): 800 x 800, yaws 0 90 180 270, pitches 30 .. 150 -- the pole views included --
    on one image, on one context and shared out to three contexts of the device (rows of every view per context)."""
    pano = synth.synth_pano(4096, 2048, 1003, "N")
    one = tmp_path / "one"
    one.mkdir()
    _save(one / "d.png", pano)
    r = _run(["--input_path", str(one), "--output_path", str(tmp_path / "o"), "--exact"], tmp_path)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    _check_folder(tmp_path / "o", "d", pano, [0, 90, 180, 270], [30, 60, 90, 120, 150], 800, 800, threaded=True)
    r = _run(["--input_path", str(one), "--output_path", str(tmp_path / "o3"), "--exact", "--devices", "0", "0", "0"], tmp_path)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    for n in sorted(os.listdir(tmp_path / "o")):
        assert np.array_equal(_load(tmp_path / "o" / n), _load(tmp_path / "o3" / n)), n


def test_cli_exact_folder_pipeline_noise(tmp_path, synth):
    """A folder of images through the two-slot device pipeline: the maps go up with a slot's job, once."""
    src = tmp_path / "in"
    src.mkdir()
    panos = [synth.synth_pano(1024, 512, 1010 + i, "N") for i in range(4)]
    for i, p in enumerate(panos):
        _save(src / ("p%d.png" % i), p)
    r = _run(["--input_path", str(src), "--output_path", str(tmp_path / "o"), "--exact", "--yaw_angles", "0", "45", "200",
              "--pitch_angles", "20", "90", "--output_width", "322", "--output_height", "181", "--FOV", "100"], tmp_path)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    for i, p in enumerate(panos):
        _check_folder(tmp_path / "o", "p%d" % i, p, [0, 45, 200], [20, 90], 322, 181, fov=100)


def test_cli_exact_refuses_float_quality(tmp_path):
    r = _run(["--input_path", str(tmp_path), "--exact", "--quality", "f16"], tmp_path)
    assert r.returncode == 2 and "--exact" in r.stderr


def test_cli_quality_flag(tmp_path, synth):
    """--quality f16 / f32 reach the float pixel path: within 2 levels of the default path on a smooth panorama."""
    pano = synth.synth_pano(1024, 512, 1001, "S")
    _save(tmp_path / "q.png", pano)
    outs = {}
    for q in ("u8", "f32", "f16"):
        r = _run(["--input_path", str(tmp_path / "q.png"), "--output_path", str(tmp_path / q), "--quality", q, "--yaw_angles", "10",
                  "--pitch_angles", "75", "--output_width", "320", "--output_height", "200"], tmp_path)
        assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
        outs[q] = _load(tmp_path / q / "q_320x200_yaw_10_pitch_75.png").astype(np.int16)
    for q in ("f32", "f16"):
        d = np.abs(outs[q][:, 1:-1] - outs["u8"][:, 1:-1])
        assert d.max() <= 2 and (outs[q] != outs["u8"]).any(), (q, int(d.max()))
    # --pixel_centres: the other sampling convention of the float paths (half a pixel), refused with the reference's arithmetic
    r = _run(["--input_path", str(tmp_path / "q.png"), "--output_path", str(tmp_path / "c"), "--quality", "f32", "--pixel_centres",
              "--yaw_angles", "10", "--pitch_angles", "75", "--output_width", "320", "--output_height", "200"], tmp_path)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    c = _load(tmp_path / "c" / "q_320x200_yaw_10_pitch_75.png").astype(np.int16)
    assert (c != outs["f32"]).mean() > 0.03 and np.abs(c - outs["f32"])[:, 2:-2].max() <= 12   # a smooth image, half a pixel apart
    r = _run(["--input_path", str(tmp_path / "q.png"), "--pixel_centres"], tmp_path)
    assert r.returncode == 2 and "--pixel_centres" in r.stderr


def test_api_exact_from_the_reference_fanout(pkg, synth):
    """process_yaw_and_pitchs(..., exact=True) called the way process_single_image calls it (P:252-265): one task per yaw
    on a thread pool, one shared noise panorama; every slice is the oracle's; the named maps are uploaded once per slot."""
    from concurrent.futures import ThreadPoolExecutor
    tool = pkg.panorama_to_plane_pitch
    pano = synth.synth_pano(2048, 1024, 1004, "N")
    yaws, pitches, ow, oh = [0, 33, 90, 181, 270, 359], [45, 90, 135], 400, 300
    with ThreadPoolExecutor(max_workers=6) as ex:
        got = list(ex.map(lambda y: tool.process_yaw_and_pitchs(pano, y, pitches, ow, oh, 90, exact=True), yaws))
    want = oracle_views(pano, yaws, pitches, ow, oh, 90)
    for yi in range(len(yaws)):
        for pi in range(len(pitches)):
            assert np.array_equal(got[yi][pi], want[yi, pi]), (yaws[yi], pitches[pi])
    # the module switch: set_exact(True) makes the plain signature exact, and get_pitch_mapping hands out the host maps
    tool.set_exact(True)
    try:
        sl = tool.process_yaw_and_pitchs(pano, 33, pitches, ow, oh)
        assert all(np.array_equal(sl[pi], want[1, pi]) for pi in range(len(pitches)))
        U, V = tool.get_pitch_mapping(ow, oh, 45, 2048, 1024, 90)
        from oracle import maps
        oU, oV = maps.pitch_map_deg(ow, oh, 45, 2048, 1024, 90)
        assert np.array_equal(U, oU, equal_nan=True) and np.array_equal(V, oV, equal_nan=True)
        with pytest.raises(ValueError):
            tool.set_quality("f16")
    finally:
        tool.set_exact(False)


def test_exact_mode_over_a_random_sequence_of_geometries(pkg, synth):
    """One process, the module's exact mode, forty calls whose geometry changes from call to call and comes back: view sizes
    (odd ones too), FOVs, pitch lists that share angles, real-valued yaws, three panorama sizes, pole pitches -- the one-shot
    slots keep maps under the names _exact_maps gives them, the contexts keep plans by geometry, and every view is the
    oracle's bytes on noise whatever the call before it was."""
    tool = pkg.panorama_to_plane_pitch
    rng = np.random.default_rng(20260604)
    panos = {pw: synth.synth_pano(pw, pw // 2, 1100 + pw, "N") for pw in (512, 1024, 1536)}
    geos = []
    for _ in range(8):
        ow, oh = int(rng.integers(17, 260)), int(rng.integers(17, 200))
        n_p = int(rng.integers(1, 4))
        pitches = sorted(set(int(x) for x in rng.choice([1, 20, 45, 60, 90, 120, 150, 179], n_p, replace=False)))
        geos.append((ow, oh, pitches, int(rng.choice([60, 90, 110])), int(rng.choice(list(panos)))))
    tool.set_exact(True)
    try:
        for call in range(40):
            gi = int(rng.integers(0, len(geos)))
            ow, oh, pitches, fov, pw = geos[gi]
            yaw = float(np.round(rng.uniform(-30, 400), 1)) if call % 3 else int(rng.integers(0, 360))
            got = tool.process_yaw_and_pitchs(panos[pw], yaw, pitches, ow, oh, fov)
            want = oracle_views(panos[pw], [yaw], pitches, ow, oh, fov)[0]
            for pi in range(len(pitches)):
                assert np.array_equal(got[pi], want[pi]), (call, gi, yaw, pitches[pi], ow, oh, fov, pw)
    finally:
        tool.set_exact(False)


def test_exact_mode_on_the_seam_row_and_the_pole_pixel(pkg, synth):
    """The geometries where device-evaluated maps are KNOWN to differ from NumPy's (tests/test_gpu_fused_exceptions.py: an
    output row whose rays land exactly on the seam, FOV 120 / pitch 30 / W = 474; the pixel that looks a fraction of a row
    past a pole): with the host-evaluated maps of the exact mode the views are the oracle's bytes there too, on noise."""
    tool = pkg.panorama_to_plane_pitch
    for pw, ow, oh, fov, yaws, pitches, seed in ((1024, 474, 344, 120, [297, 314], [30, 129], 735),
                                                (1024, 372, 183, 90, [285, 252], [9, 54], 825)):
        pano = synth.synth_pano(pw, pw // 2, seed, "N")
        got = tool.process_views(pano, yaws, pitches, ow, oh, fov, exact=True)
        assert np.array_equal(got, oracle_views(pano, yaws, pitches, ow, oh, fov)), (pw, ow, oh, fov)


def test_exact_maps_through_the_view_sharded_driver(pkg, synth):
    """One image shared out to several contexts by WHOLE VIEWS (a masked job per context, its pitch subset of the maps) and
    by rows, with the exact mode's maps: the oracle's bytes either way."""
    import importlib
    drv = importlib.import_module("360-to-planer-images_amd._driver")
    em = importlib.import_module("360-to-planer-images_amd._exact_maps")
    pano = synth.synth_pano(1024, 512, 1005, "N")
    yaws, pitches, ow, oh, fov = [0, 45, 90, 200], [50, 90, 130], 160, 120, 90
    maps = em.pitch_map_stack(ow, oh, pitches, 1024, 512, fov)
    want = oracle_views(pano, yaws, pitches, ow, oh, fov)
    try:
        for how in ("views", "rows"):
            got = drv.process_views_sharded(pano, [float(y) for y in yaws], [float(p) for p in pitches], ow, oh, float(fov),
                                            [0, 0, 0], how=how, maps=maps)
            assert np.array_equal(got, want), how
    finally:
        drv.release_sharded()
        em.clear()


def test_pitch_maps_entry_point_keys_and_yaws(gpu, synth):
    """p2p_remap_views_pitch_maps_f64 directly: without a key (maps uploaded and planned per call), with a key (the slot keeps
    them: other panoramas, other yaw VALUES through the same maps), and a second key with other maps -- always the bytes of
    the all-caller-maps entry point fed the oracle's yaw rows."""
    from tests._util import oracle_maps
    pw, ph, ow, oh, fov = 1024, 512, 222, 148, 80
    pitches = [40, 90, 141]
    panos = [synth.synth_pano(pw, ph, 1020 + i, "N") for i in range(2)]

    def want(pano, yaws, U, V):
        rows, _, _ = oracle_maps(yaws, pitches, ow, oh, pw, ph, fov)
        return gpu.remap_views_maps(pano, rows, U, V)

    _, U, V = oracle_maps([0], pitches, ow, oh, pw, ph, fov)
    _, U2, V2 = oracle_maps([0], [p + 3 for p in pitches], ow, oh, pw, ph, fov)
    for key in (0, 4242):
        for pano in panos:
            for yaws in ([0, 33, 270], [10, 200, 359]):   # same count: the slot's job is re-used, its yaw tables rebuilt
                got = gpu.remap_views_pitch_maps(pano, yaws, U, V, key)
                assert np.array_equal(got, want(pano, yaws, U, V)), (key, yaws)
    got = gpu.remap_views_pitch_maps(panos[0], [0, 33, 270], U2, V2, 4243)     # other maps under another name
    assert np.array_equal(got, want(panos[0], [0, 33, 270], U2, V2))
    got = gpu.remap_views_pitch_maps(panos[1], [0, 33, 270], U, V, 4242)       # and the first name still means the first maps
    assert np.array_equal(got, want(panos[1], [0, 33, 270], U, V))
    with pytest.raises(ValueError):
        gpu.remap_views_pitch_maps(panos[0], [0], U, V[:2], 0)
    gpu.release_cache()
