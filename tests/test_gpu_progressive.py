"""Progressive / resumable accumulation and the command-line renderer, on the GPU."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO
from test_scenefile import _teapot_doc

pytestmark = pytest.mark.gpu


def test_progressive_equals_one_shot_and_resumes(pt, scenes, tmp_path):
    from ptamd.progressive import Accumulator

    sc, cam = pt.scene_from_description(scenes.teapot_scene(res=(128, 72)))
    one = pt.render_image(sc, cam, 64, seed=7)
    acc = Accumulator(sc, cam, seed=7)
    acc.add(32)
    acc.save(str(tmp_path / "ck.npz"))
    acc.add(32)
    two = acc.radiance_sum()
    # the same 64 samples per pixel, summed in another order
    assert np.abs(two - one).max() <= 1e-5 * max(1.0, float(np.abs(one).max()))
    resumed = Accumulator(sc, cam, seed=7)
    resumed.load(str(tmp_path / "ck.npz"))
    assert resumed.samples_done == 32
    resumed.add(32)
    assert np.array_equal(resumed.radiance_sum(), two)          # a resumed render is bit-identical
    img = resumed.image()
    assert img.shape == (72, 128, 3) and img.min() >= 0 and img.max() <= 1.0 + 1e-6
    wrong = Accumulator(sc, cam, seed=8)
    with pytest.raises(ValueError, match="another seed"):
        wrong.load(str(tmp_path / "ck.npz"))


def test_cli_renders_scene_file(pt, scenes, tmp_path):
    from ptamd import scenefile as sf

    path = tmp_path / "teapot.json"
    path.write_text(json.dumps(_teapot_doc(tmp_path, res=(96, 54))))
    env = dict(os.environ, PYTHONPATH=os.path.join(REPO, "cuda-path-tracer_amd", "python"))
    out, hdr, ck = tmp_path / "o.ppm", tmp_path / "o.pfm", tmp_path / "ck.npz"
    r = subprocess.run([sys.executable, "-m", "ptamd.cli", str(path), "--out", str(out), "--hdr-out", str(hdr), "--chunk", "32",
                        "--checkpoint", str(ck)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    desc, settings = sf.load_scene(str(path))
    sc, cam = pt.scene_from_description(desc)
    from ptamd.progressive import Accumulator

    acc = Accumulator(sc, cam, seed=settings["seed"])
    acc.add(32)
    acc.add(32)
    from ptamd import io as ptio

    want = ptio.to_ppm_bytes(acc.image())
    data = open(out, "rb").read()
    body = data.split(b"255\n", 1)[1]
    assert np.array_equal(np.frombuffer(body, np.uint8).reshape(54, 96, 3), want)
    assert os.path.getsize(hdr) > 54 * 96 * 12 and os.path.exists(ck)
    # resuming a finished render adds nothing and writes the same picture
    r = subprocess.run([sys.executable, "-m", "ptamd.cli", str(path), "--out", str(tmp_path / "o2.ppm"), "--chunk", "32",
                        "--checkpoint", str(ck), "--resume"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "resumed at 64" in r.stderr
    assert open(tmp_path / "o2.ppm", "rb").read() == data
