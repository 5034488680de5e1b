"""The host C++ API (cuda-path-tracer_amd/host/path_tracer.h, the drop-in for the reference's class
API used by src/main.cu) checked through its command-line driver `facade_check`: mesh loading and
transforms against the Python scene library (bit-exact), the file readers against the writers of
ptamd.export_assets, and -- on the GPU -- a render through Renderer::render against the same scene
rendered through the ctypes binding (bit-exact: same library, same inputs)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import REPO

LIBDIR = os.path.join(REPO, "cuda-path-tracer_amd", "lib")
CHECK = os.path.join(LIBDIR, "facade_check")
TEAPOT = os.path.join(REPO, "examples", "teapot", "utah.obj")
FLOOR = os.path.join(REPO, "examples", "cornellbox", "floor.obj")


def _run(*args, timeout=300):
    assert os.path.exists(CHECK), "build the host library first: make -C cuda-path-tracer_amd/host"
    return subprocess.run([CHECK] + [str(a) for a in args], capture_output=True, text=True, timeout=timeout)


def _read_mesh(path):
    raw = open(path, "rb").read()
    nv, nt, nuv = np.frombuffer(raw, np.int32, 3)
    off = 12
    verts = np.frombuffer(raw, np.float32, nv * 6, off).reshape(nv, 6)
    off += nv * 24
    tri = np.frombuffer(raw, np.int32, nt * 6, off).reshape(nt, 6)
    off += nt * 24
    uvs = np.frombuffer(raw, np.float32, nuv * 2, off).reshape(nuv, 2)
    off += nuv * 8
    tail = np.frombuffer(raw, np.float32, 9, off)
    return verts[:, :3], verts[:, 3:], tri, uvs, tail[:3], tail[3:6], tail[6:]


def _face_normal_average(pos, tri):
    """src/entity.cu:110-133 in fp32, triangle by triangle in file order"""
    f = np.float32
    n = np.zeros_like(pos)
    for a, b, c in tri[:, :3]:
        e1, e2 = pos[b] - pos[a], pos[c] - pos[a]
        cr = np.array([e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]], dtype=f)
        r = f(1.0) / np.sqrt(f(f(cr[0] * cr[0] + cr[1] * cr[1]) + cr[2] * cr[2]), dtype=f)
        t = (cr * r).astype(f)
        n[a] += t
        n[b] += t
        n[c] += t
    s = (n[:, 0] * n[:, 0] + n[:, 1] * n[:, 1]).astype(f) + n[:, 2] * n[:, 2]
    return (n * (f(1.0) / np.sqrt(s, dtype=f))[:, None]).astype(f)


def test_entity_obj_loading_and_normals(scenes, tmp_path):
    out = tmp_path / "m.bin"
    r = _run("mesh", TEAPOT, out)
    assert r.returncode == 0, r.stderr
    pos, nrm, tri, uvs, bmin, bmax, center = _read_mesh(out)
    ppos, ptri, puv, _ = scenes.load_obj(TEAPOT)
    assert np.array_equal(pos, ppos) and np.array_equal(tri, ptri)
    assert len(uvs) == len(puv)
    with np.errstate(invalid="ignore", divide="ignore"):
        want = _face_normal_average(ppos, ptri)
    assert np.array_equal(nrm, want, equal_nan=True)
    pad = np.float32(0.0001)
    assert np.array_equal(bmin, ppos.min(axis=0) - pad) and np.array_equal(bmax, ppos.max(axis=0) + pad)
    assert np.array_equal(center, scenes.mesh_center(ppos))


def test_entity_transforms_match_scene_library(scenes, tmp_path):
    out = tmp_path / "m.bin"
    rot = (-3.14159265 / 2.0, 3.14159265 / 2.0, 0.3)
    r = _run("mesh", TEAPOT, out, "s1.75", "r%.9g,%.9g,%.9g" % tuple(np.float32(rot)), "t0,22.65,-3.5")
    assert r.returncode == 0, r.stderr
    pos, nrm, tri, uvs, bmin, bmax, center = _read_mesh(out)
    ppos, ptri, _, _ = scenes.load_obj(TEAPOT)
    c = scenes.mesh_center(ppos)
    p = scenes.scale_mesh(ppos, c, 1.75)
    with np.errstate(invalid="ignore", divide="ignore"):
        n0 = _face_normal_average(ppos, ptri)
    p, n = scenes.rotate_mesh(p, n0, c, rot)
    p, c = scenes.translate_mesh(p, c, (0, 22.65, -3.5))
    assert np.array_equal(pos, p)
    assert np.array_equal(nrm, n, equal_nan=True)
    assert np.array_equal(center, c)
    pad = np.float32(0.0001)
    assert np.array_equal(bmin, p.min(axis=0) - pad) and np.array_equal(bmax, p.max(axis=0) + pad)


def test_entity_rejects_bad_files(tmp_path):
    r = _run("mesh", tmp_path / "missing.obj", tmp_path / "o.bin")
    assert r.returncode != 0 and "cannot load" in r.stdout
    quad = tmp_path / "quad.obj"
    quad.write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf 1 2 3 4\n")
    r = _run("mesh", quad, tmp_path / "o.bin")
    assert r.returncode != 0 and "must be triangles" in r.stderr


def test_hdr_reader_flat_and_rle(tmp_path):
    from ptamd import export_assets as ea

    rng = np.random.default_rng(5)
    img = (rng.random((24, 40, 3), dtype=np.float32) * np.float32(30.0)).astype(np.float32)
    img[3, 5] = 0.0
    img[7, 9] = (1e-3, 2e-3, 5e-4)
    for rle in (False, True):
        path = tmp_path / ("a%d.hdr" % rle)
        rgbe = ea.write_hdr(str(path), img, rle=rle)
        out = tmp_path / "h.bin"
        r = _run("hdr", path, out)
        assert r.returncode == 0, r.stderr
        raw = open(out, "rb").read()
        w, h = np.frombuffer(raw, np.int32, 2)
        assert (w, h) == (40, 24)
        got = np.frombuffer(raw, np.float32, w * h * 3, 8).reshape(h, w, 3)
        assert np.array_equal(got, ea.rgbe_decode(rgbe))
        assert np.abs(got - img).max() <= img.max() / 128.0     # 8-bit shared-exponent mantissas
    assert _run("hdr", tmp_path / "nope.hdr", tmp_path / "h.bin").returncode != 0


def test_texture_readers(tmp_path):
    from ptamd import export_assets as ea

    rng = np.random.default_rng(6)
    rgb = rng.integers(0, 256, (9, 13, 3), dtype=np.uint8)
    gray = rng.integers(0, 256, (9, 13), dtype=np.uint8)
    ea.write_pnm(str(tmp_path / "c.ppm"), rgb)
    ea.write_pnm(str(tmp_path / "g.pgm"), gray)
    for name, ch, want in (("c.ppm", 3, rgb), ("g.pgm", 1, gray), ("c.ppm", 1, rgb[..., 0]), ("g.pgm", 3, np.repeat(gray[..., None], 3, 2))):
        out = tmp_path / "t.bin"
        r = _run("pnm", tmp_path / name, ch, out)
        assert r.returncode == 0, r.stderr
        raw = open(out, "rb").read()
        w, h = np.frombuffer(raw, np.int32, 2)
        assert (w, h) == (13, 9)
        assert np.array_equal(np.frombuffer(raw, np.uint8, w * h * ch, 8).reshape(want.shape), want)
    (tmp_path / "bad.ppm").write_bytes(b"P3\n1 1\n255\n0 0 0\n")
    assert _run("pnm", tmp_path / "bad.ppm", 3, tmp_path / "t.bin").returncode != 0     # texture stays unset


def test_exported_obj_round_trips(scenes, tmp_path):
    from ptamd import export_assets as ea

    pos, faces, uv = scenes.statue_mesh(2000)
    path = tmp_path / "s.obj"
    ea.write_obj(str(path), pos, scenes.tri6(faces, faces), uv)
    out = tmp_path / "m.bin"
    r = _run("mesh", path, out)
    assert r.returncode == 0, r.stderr
    gpos, _, gtri, guv, _, _, _ = _read_mesh(out)
    assert np.array_equal(gpos, pos) and np.array_equal(gtri, scenes.tri6(faces, faces)) and np.array_equal(guv, uv)
    ppos, ptri, puv, _ = scenes.load_obj(str(path))
    assert np.array_equal(ppos, pos) and np.array_equal(ptri, gtri) and np.array_equal(puv, uv)


@pytest.mark.gpu
def test_renderer_render_matches_binding(pt, scenes, tmp_path):
    w, h, spp = 160, 90, 64
    out = tmp_path / "r.bin"
    r = _run("teapot", TEAPOT, FLOOR, w, h, spp, out)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "start render" in r.stdout and "Device 0" in r.stdout
    raw = open(out, "rb").read()
    gw, gh, gspp = np.frombuffer(raw, np.int32, 3)
    assert (gw, gh, gspp) == (w, h, spp)
    n = w * h * 3
    hdr_sum = np.frombuffer(raw, np.float32, n, 12).reshape(h, w, 3)
    ldr = np.frombuffer(raw, np.float32, n, 12 + 4 * n).reshape(h, w, 3)
    sc, cam = pt.scene_from_description(scenes.teapot_scene(res=(w, h)))
    want_sum = pt.render_image(sc, cam, spp, seed=1337, n_split=4)
    assert np.array_equal(hdr_sum, want_sum)
    assert np.array_equal(ldr, pt.post_chain_host(want_sum, spp))
    ppm = open(str(out) + ".ppm", "rb").read()
    assert ppm.startswith(b"P6\n") and len(ppm.split(b"255\n", 1)[1]) == w * h * 3


def test_runtime_octree_depth_through_the_class_api(pt):
    """Entity::construct_octree(depth): the default is the reference's fixed depth 6 (known node counts of the
    teapot), any depth up to 10 is accepted, and the tree equals the one the binding builds."""
    r = _run("octree", TEAPOT, 6)
    assert r.returncode == 0, r.stderr
    assert r.stdout.split() == "depth 6 nodes 24734 leaf_refs 47874 max_leaf 32 per_depth 1 8 52 226 988 4289 19170".split()
    r = _run("octree", TEAPOT, 9)
    assert r.returncode == 0, r.stderr
    from ptamd import scenes as S

    pos, tri, _, _ = S.load_obj(TEAPOT)
    sc = pt.Scene(-1)
    info = sc.octree_info(sc.add_mesh(pos, tri, sc.add_material(), octree_depth=9))
    words = r.stdout.split()
    assert int(words[1]) == 9 and int(words[3]) == info["n_nodes"] and int(words[5]) == info["n_leaf_refs"]
    assert [int(x) for x in words[9:]] == info["nodes_per_depth"].tolist()
    r = _run("octree", TEAPOT, 11)
    assert r.returncode != 0 and "depth must be in" in r.stderr


@pytest.mark.parametrize("depth", [3, 6])
def test_pointer_octree_class_equals_the_flat_tree(pt, tmp_path, depth):
    """The Octree class of the host API (the reference's pointer tree, src/geometry.cuh:93-133, filled triangle by
    triangle with insert_triangles as src/entity.cu:167-170 does) against the library's level-by-level builder:
    the same nodes, child slots and leaf triangle lists in the same order; AABB::intersects_triangle answers."""
    from ptamd import scenes as S

    out = tmp_path / "ptree.bin"
    r = _run("ptree", TEAPOT, depth, out)
    assert r.returncode == 0, r.stderr
    words = np.fromfile(out, dtype=np.int32)
    pos, tri, _, _ = S.load_obj(TEAPOT)
    sc = pt.Scene(-1)
    flat = sc.octree_dump(sc.add_mesh(pos, tri, sc.add_material(), octree_depth=depth))
    assert words.shape == flat.shape and (words == flat).all()
    assert r.stdout.split() == ["words", str(len(flat)), "overlap", "1", "miss", "0"]


@pytest.mark.gpu
def test_renderer_depth_nine_and_two_device_render(pt, scenes, tmp_path):
    """Renderer::render with an octree of depth 9 requested through the class API, on one GPU and as a
    two-"device" render (both entries name GPU 0: two host threads, two scene copies, two sample shares,
    peer-copy sum): the radiance sums equal the binding's render of the same scene at depth 9 (bit-exact on
    one device; equal up to the association of the two shares' sums on two)."""
    w, h, spp = 128, 72, 64
    desc = scenes.teapot_scene(res=(w, h), octree_depth=9)
    sc, cam = pt.scene_from_description(desc)
    want = pt.render_image(sc, cam, spp, seed=1337, n_split=4)
    lo = pt.render_image(sc, cam, spp // 2, seed=1337, n_split=4)
    hi = pt.render_image(sc, cam, spp // 2, seed=1337, n_split=4, sample_begin=spp // 2)
    sc.close()
    n = w * h * 3
    for devices, expect in ((None, want), ("0,0", lo + hi)):
        out = tmp_path / "r.bin"
        args = ["teapot", TEAPOT, FLOOR, w, h, spp, out, 9] + ([devices] if devices else [])
        r = _run(*args)
        assert r.returncode == 0, r.stdout + r.stderr
        raw = open(out, "rb").read()
        hdr_sum = np.frombuffer(raw, np.float32, n, 12).reshape(h, w, 3)
        ldr = np.frombuffer(raw, np.float32, n, 12 + 4 * n).reshape(h, w, 3)
        assert np.array_equal(hdr_sum, expect), devices
        assert np.array_equal(ldr, pt.post_chain_host(expect, spp)), devices
        # the teapot's octree is built once however many GPUs render (the second device gets a clone of the built scene)
        assert "octrees built by the library: 1\n" in r.stdout, r.stdout
        if devices:
            assert r.stdout.count("Device 0") == 2


@pytest.mark.gpu
def test_post_functions_match_fused_chain(pt, tmp_path):
    """normalize / threshold / gaussian bloom / add / ACES / gamma / bilateral called one by one through
    the class API equal the device-resident chain the renderer uses"""
    w, h = 96, 64
    out = tmp_path / "p.bin"
    r = _run("post", w, h, out)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = open(out, "rb").read()
    n = w * h * 3
    src = np.frombuffer(raw, np.float32, n, 8).reshape(h, w, 3)
    got = np.frombuffer(raw, np.float32, n, 8 + 4 * n).reshape(h, w, 3)
    assert np.array_equal(got, pt.post_chain_host(src, 32))
