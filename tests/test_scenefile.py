"""Scene-file front end (ptamd.scenefile): file readers against the writers of ptamd.export_assets,
and a scene file against the same scene assembled by hand with the scene library."""
import json
import os

import numpy as np
import pytest

from conftest import REPO

TEAPOT = os.path.join(REPO, "examples", "teapot", "utah.obj")
FLOOR = os.path.join(REPO, "examples", "cornellbox", "floor.obj")


def _teapot_doc(tmp_path, res=(96, 54)):
    return {
        "materials": {"floor": {"albedo": [0.8, 0.8, 0.8]},
                      "pot": {"albedo": [0.8, 0.6, 0.3], "specular": [0.3, 0.3, 0.3], "smoothness": 0.5, "smooth_shading": True},
                      "light": {"albedo": [1, 1, 1], "emission": 10.0}},
        "entities": [{"obj": TEAPOT, "material": "pot", "octree": True},
                     {"obj": FLOOR, "material": "floor",
                      "transforms": [{"scale": 20.0}, {"translate": [-205.3 - 27.5, -36.4, 82.5 - 28.5]}]},
                     {"sphere": {"center": [-150.0, 160.0, -40.0], "radius": 40.0}, "material": "light"}],
        "camera": {"position": [-205.3, 40.0, -150.0], "direction": [0, -0.12, 1], "resolution": list(res), "focal_length": 1.6},
        "render": {"spp": 64, "seed": 1337},
    }


def test_readers_round_trip(tmp_path):
    from ptamd import export_assets as ea
    from ptamd import scenefile as sf

    rng = np.random.default_rng(1)
    rgb = rng.integers(0, 256, (7, 11, 3), dtype=np.uint8)
    gray = rng.integers(0, 256, (7, 11), dtype=np.uint8)
    ea.write_pnm(str(tmp_path / "c.ppm"), rgb)
    ea.write_pnm(str(tmp_path / "g.pgm"), gray)
    assert np.array_equal(sf.read_pnm(str(tmp_path / "c.ppm")), rgb)
    assert np.array_equal(sf.read_pnm(str(tmp_path / "g.pgm")), gray)
    img = (rng.random((12, 20, 3), dtype=np.float32) * np.float32(50)).astype(np.float32)
    for rle in (False, True):
        rgbe = ea.write_hdr(str(tmp_path / "e.hdr"), img, rle=rle)
        assert np.array_equal(sf.read_hdr(str(tmp_path / "e.hdr")), ea.rgbe_decode(rgbe))
    (tmp_path / "bad.ppm").write_bytes(b"P3\n1 1\n255\n0 0 0\n")
    with pytest.raises(ValueError):
        sf.read_pnm(str(tmp_path / "bad.ppm"))


def test_scene_file_equals_hand_built_scene(tmp_path, scenes):
    from ptamd import scenefile as sf

    path = tmp_path / "teapot.json"
    path.write_text(json.dumps(_teapot_doc(tmp_path)))
    desc, settings = sf.load_scene(str(path))
    want = scenes.teapot_scene(res=(96, 54))
    assert settings == dict(spp=64, seed=1337)
    assert desc["camera"] == want["camera"]
    got_by_kind = [e["kind"] for e in desc["entities"]]
    assert got_by_kind == ["mesh", "mesh", "sphere"]
    for g, w in zip(desc["entities"], want["entities"]):
        if g["kind"] == "mesh":
            assert np.array_equal(g["positions"], w["positions"]) and np.array_equal(g["tri"], w["tri"])
            assert g["octree_depth"] == w["octree_depth"] and g["normals"] is None
        else:
            assert g["center"] == w["center"] and g["radius"] == w["radius"]
    # same materials, referenced by name instead of position
    for g, w in zip(desc["entities"], want["entities"]):
        gm, wm = desc["materials"][g["material"]], want["materials"][w["material"]]
        assert {k: gm[k] for k in ("albedo", "specular", "emission", "smoothness", "smooth_shading")} == \
               {k: wm[k] for k in ("albedo", "specular", "emission", "smoothness", "smooth_shading")}


def test_normal_modes_and_rotation(tmp_path, scenes):
    from ptamd import scenefile as sf

    obj = tmp_path / "tri.obj"
    obj.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\nvn 0 0 1\nvn 0 1 0\nvn 1 0 0\nvn 0 0 -1\n"
                   "f 1//1 2//1 3//1\nf 1//2 4//2 2//2\n")
    doc = {"materials": {"m": {}}, "camera": {"position": [0, 0, -5], "direction": [0, 0, 1], "resolution": [8, 8]},
           "entities": [{"obj": str(obj), "material": "m", "normals": "file"}]}
    p = tmp_path / "s.json"
    p.write_text(json.dumps(doc))
    n = sf.load_scene(str(p))[0]["entities"][0]["normals"]
    s = np.float32(1.0) / np.sqrt(np.float32(2.0))
    assert np.allclose(n[0], [0, s, s]) and np.allclose(n[2], [0, 0, 1]) and np.allclose(n[3], [0, 1, 0])
    doc["entities"][0]["normals"] = "reference"       # vn taken by vertex id
    p.write_text(json.dumps(doc))
    n = sf.load_scene(str(p))[0]["entities"][0]["normals"]
    assert np.array_equal(n, np.array([[0, 0, 1], [0, 1, 0], [1, 0, 0], [0, 0, -1]], np.float32))
    doc["entities"][0]["normals"] = "computed"
    doc["entities"][0]["transforms"] = [{"rotate": [0.3, -0.2, 0.1]}]
    p.write_text(json.dumps(doc))
    e = sf.load_scene(str(p))[0]["entities"][0]
    pos, tri, _, _ = scenes.load_obj(str(obj))
    c = scenes.mesh_center(pos)
    wp, wn = scenes.rotate_mesh(pos, sf.face_normal_average(pos, tri), c, (0.3, -0.2, 0.1))
    assert np.array_equal(e["positions"], wp) and np.array_equal(e["normals"], wn, equal_nan=True)
    doc["entities"][0]["material"] = "nope"
    p.write_text(json.dumps(doc))
    with pytest.raises(ValueError, match="unknown material"):
        sf.load_scene(str(p))


def test_scene_file_feeds_the_oracle(tmp_path, scenes, oracle):
    from ptamd import scenefile as sf

    path = tmp_path / "teapot.json"
    path.write_text(json.dumps(_teapot_doc(tmp_path, res=(32, 18))))
    desc, _ = sf.load_scene(str(path))
    sc, cam = oracle.scene_from_description(desc)
    rays = oracle.camera_rays(cam, 1337, 0)
    hits, _, _ = sc.trace(rays)
    sc2, cam2 = oracle.scene_from_description(scenes.teapot_scene(res=(32, 18)))
    hits2, _, _ = sc2.trace(oracle.camera_rays(cam2, 1337, 0))
    assert np.array_equal(hits["triangle"], hits2["triangle"]) and np.array_equal(hits["t"], hits2["t"])
    assert (hits["entity"] >= 0).sum() > 50
