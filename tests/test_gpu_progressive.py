"""Progressive / resumable accumulation and the command-line renderer, on the GPU."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO
from parity import assert_render_parity
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


def _assert_matches_oracle(img, c, cam_c, spp):
    """img: normalised radiance of `spp` samples per pixel from the HIP path, against the oracle's render of
    the same samples (criteria: tests/parity.py)."""
    ref, _ = c.render(cam_c, 1337, 0, spp, n_threads=8)
    conv, _ = c.render(cam_c, 1337, 1000, 1000 + 16 * spp, n_threads=8)
    assert_render_parity(img, ref / spp, conv / (16 * spp))


def test_chunked_and_resumed_accumulation_matches_oracle(pt, oracle, scenes, tmp_path):
    """64 samples per pixel added as 24 + (checkpoint, new accumulator, resume) 8 + 32: the radiance the
    accumulator reports is the oracle's render of samples 0..63."""
    from ptamd.progressive import Accumulator

    desc = scenes.teapot_scene(res=(96, 54))
    sc, cam = pt.scene_from_description(desc)
    c, cam_c = oracle.scene_from_description(desc)
    first = Accumulator(sc, cam, seed=1337, tag="t")
    first.add(24)
    ck = str(tmp_path / "state")                     # no .npz suffix: written to exactly this path
    first.save(ck)
    assert os.path.exists(ck) and not os.path.exists(ck + ".npz")
    first.close()
    acc = Accumulator(sc, cam, seed=1337, tag="t")
    acc.load(ck, max_samples=64)
    assert acc.samples_done == 24
    acc.add(8)
    acc.add(32)
    assert acc.samples_done == 64
    _assert_matches_oracle(acc.radiance_sum() / 64, c, cam_c, 64)
    with pytest.raises(ValueError, match="more than"):
        Accumulator(sc, cam, seed=1337, tag="t").load(ck, max_samples=16)
    with pytest.raises(ValueError, match="another scene"):
        Accumulator(sc, cam, seed=1337, tag="other").load(ck)
    with pytest.raises(ValueError, match="another sample index"):
        Accumulator(sc, cam, seed=1337, tag="t", first_sample=32).load(ck)
    # a shard that starts at sample 32 renders samples 32..63: first half + second half = the whole
    lo = Accumulator(sc, cam, seed=1337)
    lo.add(32)
    hi = Accumulator(sc, cam, seed=1337, first_sample=32)
    hi.add(32)
    whole = pt.render_image(sc, cam, 64, seed=1337)
    assert np.allclose(lo.radiance_sum() + hi.radiance_sum(), whole, rtol=2e-5, atol=1e-5)
    sc.close()


def test_cli_hdr_output_matches_oracle(pt, oracle, tmp_path):
    """The command-line renderer's linear HDR output (PFM) of a scene file against the oracle's render of
    the same scene description at equal spp; bad arguments are rejected instead of looping."""
    from ptamd import io as ptio
    from ptamd import scenefile as sf

    path = tmp_path / "teapot.json"
    path.write_text(json.dumps(_teapot_doc(tmp_path, res=(96, 54))))
    env = dict(os.environ, PYTHONPATH=os.path.join(REPO, "cuda-path-tracer_amd", "python"))
    hdr, ck = tmp_path / "o.pfm", tmp_path / "ck.npz"
    base = [sys.executable, "-m", "ptamd.cli", str(path), "--out", str(tmp_path / "o.ppm"), "--hdr-out", str(hdr)]
    r = subprocess.run(base + ["--chunk", "24", "--checkpoint", str(ck)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    desc, settings = sf.load_scene(str(path))
    assert settings["spp"] == 64
    c, cam_c = oracle.scene_from_description(desc)
    _assert_matches_oracle(ptio.load_pfm(str(hdr)), c, cam_c, 64)
    r = subprocess.run(base + ["--chunk", "0"], env=env, capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "--chunk" in r.stderr
    # a checkpoint of 64 samples cannot seed a 32-spp render, nor a render of an edited scene file
    r = subprocess.run(base + ["--spp", "32", "--checkpoint", str(ck), "--resume"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "more than" in r.stderr
    doc = _teapot_doc(tmp_path, res=(96, 54))
    doc["camera"]["focal_length"] = 1.7
    path.write_text(json.dumps(doc))
    r = subprocess.run(base + ["--checkpoint", str(ck), "--resume"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "another scene" in r.stderr


def test_cli_checkpoint_is_tied_to_the_files_the_scene_names(tmp_path):
    """The checkpoint's tag hashes the loaded scene description, so a mesh file edited under an unchanged scene file
    is noticed on resume (samples of two different scenes are never mixed)."""
    import shutil

    from test_scenefile import TEAPOT

    obj = tmp_path / "pot.obj"
    shutil.copy(TEAPOT, obj)
    doc = _teapot_doc(tmp_path, res=(64, 36))
    doc["entities"][0]["obj"] = str(obj)
    doc["render"]["spp"] = 32
    path = tmp_path / "scene.json"
    path.write_text(json.dumps(doc))
    env = dict(os.environ, PYTHONPATH=os.path.join(REPO, "cuda-path-tracer_amd", "python"))
    ck = tmp_path / "ck.npz"
    base = [sys.executable, "-m", "ptamd.cli", str(path), "--out", str(tmp_path / "o.ppm"), "--checkpoint", str(ck)]
    r = subprocess.run(base, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    r = subprocess.run(base + ["--resume"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "resumed at 32" in r.stderr
    lines = obj.read_text().splitlines()
    k = next(i for i, ln in enumerate(lines) if ln.startswith("v "))
    x, y, z = (float(t) for t in lines[k].split()[1:4])
    lines[k] = f"v {x + 0.5} {y} {z}"
    obj.write_text("\n".join(lines) + "\n")
    r = subprocess.run(base + ["--resume"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "another scene" in r.stderr
