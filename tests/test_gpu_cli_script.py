"""The two tool files executed as SCRIPTS on real folders (README's command lines; INTEGRATION Option A: the drop-in for
`python app/panorama_to_plane-pitch.py --input_path ...`, P:359-488, and the legacy tool, L:283-388): output names and
formats, one image shared out to several contexts of one device, the reference's argument errors."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "360-to-planer-images_amd", "panorama_to_plane_pitch.py")
LEGACY = os.path.join(ROOT, "360-to-planer-images_amd", "panorama_to_plane.py")


def _run(args, cwd):
    return subprocess.run([sys.executable] + args, cwd=cwd, capture_output=True, text=True, timeout=600)


def test_tools_as_scripts_on_a_folder(tmp_path, synth):
    from PIL import Image
    src, one = tmp_path / "in", tmp_path / "one"
    src.mkdir(); one.mkdir()
    for i in range(3):
        Image.fromarray(synth.synth_pano(1024, 512, 1000 + i, "S")[:, :, ::-1]).save(src / ("pano%d.png" % i), compress_level=1)
    Image.fromarray(synth.synth_pano(1024, 512, 1000, "S")[:, :, ::-1]).save(one / "a.png", compress_level=1)
    size = ["--output_width", "320", "--output_height", "180"]
    r = _run([TOOL, "--input_path", str(src), "--output_path", str(tmp_path / "o1"), "--output_format", "jpg",
              "--yaw_angles", "0", "90", "--pitch_angles", "60", "90"] + size, tmp_path)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    names = sorted(os.listdir(tmp_path / "o1"))
    assert len(names) == 12 and names[0] == "pano0_320x180_yaw_0_pitch_60.jpg", names[:3]  # P:275's file names
    # one image: on one context, and shared out to three contexts of the device (rows of every view per context)
    r1 = _run([TOOL, "--input_path", str(one), "--output_path", str(tmp_path / "o2")] + size, tmp_path)
    r3 = _run([TOOL, "--input_path", str(one), "--output_path", str(tmp_path / "o3"), "--devices", "0", "0", "0"] + size, tmp_path)
    assert r1.returncode == 0 and r3.returncode == 0, (r1.stderr + r3.stderr)[-2000:]
    names = sorted(os.listdir(tmp_path / "o2"))
    assert len(names) == 20 and names == sorted(os.listdir(tmp_path / "o3"))  # the reference's defaults: 4 yaws x 5 pitches (P:412-437)
    for n in names:
        assert np.array_equal(np.asarray(Image.open(tmp_path / "o2" / n)), np.asarray(Image.open(tmp_path / "o3" / n))), n
    # the reference's argument check (P:372-375) through the script
    r = _run([TOOL, "--input_path", str(src), "--output_path", str(tmp_path / "o4"), "--pitch_angles", "0", "90"], tmp_path)
    assert r.returncode == 2 and "Pitch angle must be between 1 and 179 degrees" in r.stderr
    # the legacy tool (L:283-388)
    r = _run([LEGACY, "--input_path", str(src), "--output_path", str(tmp_path / "o5"), "--pitch", "90", "--yaw_angles", "0", "60", "120",
              "--output_width", "150", "--output_height", "200"], tmp_path)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    names = sorted(os.listdir(tmp_path / "o5"))
    assert len(names) == 9 and names[0] == "pano0_pitch90_yaw0_fov90.png", names[:3]  # L:268's file names
    assert Image.open(tmp_path / "o5" / names[0]).size == (150, 200)
