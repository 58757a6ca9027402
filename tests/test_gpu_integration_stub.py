"""INTEGRATION.md "Option C" -- the stub a reference maintainer pastes into P so that the reference keeps its own
get_yaw_mapping / get_pitch_mapping (P:42-73) and only the two cv2.remap calls move to the GPU -- executed AS
WRITTEN in the document, with maps the reference's own functions produced (tests/golden/maps_golden.npz, generator
tests/golden/make_golden_maps.py), against the CPU oracle on noise: the route that is the reference's bytes by
construction."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _stub_source():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = text.split("<!-- option-c-begin -->")[1].split("<!-- option-c-end -->")[0]
    return re.search(r"```python\n(.*?)```", block, re.S).group(1)


def test_option_c_stub_with_reference_generated_maps(gpu, golden, synth):
    from oracle import cpu_ref

    z, meta = golden
    tiny = [e for e in meta["tiny"] if "pitch" in e]
    yaws = [e for e in meta["tiny"] if "yaw" in e]
    assert tiny and yaws
    pw, ph, ow, oh = tiny[0]["pw"], tiny[0]["ph"], tiny[0]["ow"], tiny[0]["oh"]
    assert all(e["pw"] == pw for e in yaws)
    pitch_maps = {(e["fov"], e["pitch"]): (z[e["key"] + "_U"], z[e["key"] + "_V"]) for e in tiny}
    yaw_rows = {e["yaw"]: z[e["key"] + "_U"][0] for e in yaws}  # every row of the reference's U_yaw is this one

    calls = []

    def get_yaw_mapping(pano_width, pano_height, yaw_angle):   # what P:42-52 returns, from the reference's own run
        calls.append(("yaw", yaw_angle))
        U = np.ascontiguousarray(np.broadcast_to(yaw_rows[yaw_angle], (pano_height, pano_width)))
        V = np.ascontiguousarray(np.broadcast_to(np.arange(pano_height, dtype=np.float32)[:, None], (pano_height, pano_width)))
        return U, V

    def get_pitch_mapping(output_width, output_height, pitch_angle, pano_width, pano_height, fov_deg=90):  # P:55-73
        calls.append(("pitch", pitch_angle))
        return pitch_maps[(fov_deg, pitch_angle)]

    src = _stub_source().replace("/path/to/360-to-planer-images_amd/libp2p_hip.so", gpu.LIB_PATH)
    ns = {"get_yaw_mapping": get_yaw_mapping, "get_pitch_mapping": get_pitch_mapping}
    exec(compile(src, "INTEGRATION.md:option-c", "exec"), ns)
    stub = ns["process_yaw_and_pitchs"]

    pano = synth.synth_pano(pw, ph, 4242, "N")
    fovs = sorted({f for f, _ in pitch_maps})
    checked = 0
    for yaw in sorted(yaw_rows):
        for fov in fovs:
            pitches = sorted(p for f, p in pitch_maps if f == fov)
            got = stub(pano, yaw, pitches, ow, oh, fov)
            Uy, Vy = get_yaw_mapping(pw, ph, yaw)
            rot = cpu_ref.remap(pano, Uy, Vy)                          # P:192-199
            for i, p in enumerate(pitches):
                want = cpu_ref.remap(rot, *pitch_maps[(fov, p)])       # P:212-218
                assert np.array_equal(got[i], want), (yaw, fov, p)
                checked += 1
    assert checked == len(yaw_rows) * len(pitch_maps)
    assert ("yaw", sorted(yaw_rows)[0]) in calls and ("pitch", pitches[0]) in calls  # the stub went through the getters


def test_process_views_maps_argument_is_the_same_route(gpu, pkg, golden, synth):
    z, meta = golden
    tiny = [e for e in meta["tiny"] if "pitch" in e and e["fov"] == 90]
    yaws = [e for e in meta["tiny"] if "yaw" in e][:3]
    pw, ph, ow, oh = tiny[0]["pw"], tiny[0]["ph"], tiny[0]["ow"], tiny[0]["oh"]
    pano = synth.synth_pano(pw, ph, 4243, "N")
    rows = np.stack([z[e["key"] + "_U"][0] for e in yaws])
    U = np.stack([z[e["key"] + "_U"] for e in tiny])
    V = np.stack([z[e["key"] + "_V"] for e in tiny])
    a = pkg.process_views(pano, None, None, ow, oh, 90, maps=(rows, U, V))
    b = gpu.remap_views_maps(pano, rows, U, V)
    assert a.shape == (len(yaws), len(tiny), oh, ow, 3) and np.array_equal(a, b)
