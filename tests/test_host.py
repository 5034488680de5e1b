"""CPU tests of the product's host side: the C ABI exports, the octree builder and flattener
(against the oracle's pointer tree), argument checking, sample sharding."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import REPO


def test_library_exports_every_declared_symbol(pt):
    L = pt.lib()
    with open(os.path.join(REPO, "include", "ptamd.h")) as f:
        hdr = f.read()
    declared = sorted(set(re.findall(r"\b(pt_[a-z_0-9]+)\s*\(", hdr)))
    assert declared, "no declarations found"
    assert sorted(pt.ABI_SYMBOLS) == declared
    for name in declared:
        assert hasattr(L, name), name


def test_struct_layouts_match_header(pt, oracle):
    assert C.sizeof(pt.Camera) == 56 == C.sizeof(oracle.Camera)
    assert pt.HIT_DTYPE.itemsize == 52 == oracle.HIT_DTYPE.itemsize
    assert C.sizeof(pt.Counters) == 56
    assert C.sizeof(pt.MaterialDesc) == 36 + 4 + 4 * 16


def _mesh_cases(scenes, teapot):
    pos, tri = teapot
    yield "teapot-d6", pos, tri, 6
    yield "teapot-d4", pos, tri, 4
    yield "teapot-d8", pos, tri, 8
    fpos, ftri, _, _ = scenes.load_obj(os.path.join(REPO, "examples", "cornellbox", "floor.obj"))
    yield "floor-d3", fpos, ftri, 3
    spos, sfaces, suv = scenes.statue_mesh(6000)
    yield "statue6k-d6", spos, scenes.tri6(sfaces, sfaces), 6
    gpos, gfaces, guv = scenes.grid_terrain(40, 100.0, 20.0, 5)
    yield "grid-d5", gpos, scenes.tri6(gfaces, gfaces), 5
    # degenerate triangles (zero area, repeated vertices) and axis-aligned slivers
    dpos = np.array([[0, 0, 0], [1, 0, 0], [1, 0, 0], [0, 1, 0], [0.5, 0.5, 0], [0, 0, 1], [1, 1, 1]], np.float32)
    dtri = scenes.tri6(np.array([[0, 1, 2], [0, 1, 3], [4, 4, 4], [0, 5, 6], [3, 5, 1]], np.int32))
    yield "degenerate-d4", dpos, dtri, 4


def test_flat_octree_equals_reference_pointer_tree(pt, oracle, scenes, teapot):
    for name, pos, tri, depth in _mesh_cases(scenes, teapot):
        sc = pt.Scene(-1)
        e = sc.add_mesh(pos, tri, sc.add_material(), octree_depth=depth)
        osc = oracle.OracleScene()
        oe = osc.add_mesh(pos, tri, osc.add_material(), octree_depth=depth)
        assert np.array_equal(sc.octree_dump(e), osc.octree_dump(oe)), name
        info = sc.octree_info(e)
        per, refs, mx = osc.octree_stats(oe, depth)
        assert info["nodes_per_depth"].tolist() == per.tolist(), name
        assert info["n_leaf_refs"] == refs and info["max_leaf"] == mx, name
        sc.close()


def test_host_only_scene_rejects_commit_and_bad_arguments(pt, teapot):
    pos, tri = teapot
    sc = pt.Scene(-1)
    m = sc.add_material()
    with pytest.raises(pt.PtError, match="host-only"):
        sc.commit()
    with pytest.raises(pt.PtError, match="unknown material"):
        sc.add_mesh(pos, tri, 5)
    with pytest.raises(pt.PtError, match="depth out of range"):
        sc.add_mesh(pos, tri, m, octree_depth=13)
    bad = tri.copy()
    bad[7, 1] = len(pos)
    with pytest.raises(pt.PtError, match="vertex index out of range"):
        sc.add_mesh(pos, bad, m)
    with pytest.raises(pt.PtError, match="empty mesh"):
        sc.add_mesh(pos, tri[:0], m)
    with pytest.raises(pt.PtError, match="no octree"):
        e = sc.add_mesh(pos, tri, m, octree_depth=0)
        sc.octree_info(e)
    with pytest.raises(pt.PtError, match="rotation"):
        sc.set_hdri(np.zeros((2, 2, 3), np.float32), rot=-0.5)
    # an environment map of 4 GiB or more is refused before a byte of it is read (texels are addressed by 32-bit offsets)
    tiny = np.zeros((2, 2, 3), np.float32)
    with pytest.raises(pt.PtError, match="4 GiB"):
        pt.check(sc.L.pt_scene_set_hdri(sc.h, tiny.ctypes.data, 20000, 18000, 1.0, 1.0, 0.0, 0))
    sc.close()


def test_gaussian_tables_match_oracle(pt, oracle):
    for size, sigma in ((32, 10.0), (2, 0.7), (1, 3.0), (0, 1.0)):
        assert np.array_equal(pt.make_gaussian(size, sigma), oracle.make_gaussian(size, sigma))
    assert np.array_equal(pt.make_mean(3), oracle.make_mean(3))


def test_sat_and_tap_tables_match_the_committed_vectors(pt):
    """The library's host-side SAT test (what its octree builder inserts triangles with) and tap tables
    against the committed single-function vectors."""
    from conftest import GOLDEN

    t0 = np.load(os.path.join(GOLDEN, "t0_vectors.npz"))
    for i in range(len(t0["sat_bmin"])):
        for pad in (0, 1):
            got = pt.box_intersects_triangle(t0["sat_bmin"][i], t0["sat_bmax"][i], *t0["sat_tri"][i], pad_box=bool(pad))
            assert got == bool(t0["sat_result"][i, pad]), (i, pad)
    for size, sigma, key in ((2, 0.7, "gauss_2_0p7"), (32, 10.0, "gauss_32_10"), (1, 3.0, "gauss_1_3")):
        np.testing.assert_allclose(pt.make_gaussian(size, sigma), t0[key], rtol=1e-6, atol=0)
    assert np.array_equal(pt.make_mean(3), t0["mean_3"])


def test_shard_samples_partition():
    from ptamd import dist

    for n in (32, 2048, 1000, 7):
        for world in (1, 2, 4, 8):
            cover = []
            for r in range(world):
                b, e = dist.shard_samples(n, r, world)
                assert 0 <= b <= e <= n
                cover += list(range(b, e))
            assert cover == list(range(n))
    assert dist.effective_spp(2048) == 2048 and dist.effective_spp(1000) == 992 and dist.effective_spp(5) == 32
    with pytest.raises(ValueError):
        dist.shard_samples(10, 2, 2)


def test_obj_reader_conventions(scenes, tmp_path):
    p = tmp_path / "m.obj"
    p.write_text("# c\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 0 1\nvn 0 0 1\nf 1/1/1 2/2/1 3/3/1\nf -3//1 -2//1 -1//1\n")
    pos, tri, uv, vn = scenes.load_obj(str(p))
    assert pos.shape == (3, 3) and uv.shape == (3, 2) and vn.shape == (1, 3)
    assert tri.tolist() == [[0, 1, 2, 0, 1, 2], [0, 1, 2, -1, -1, -1]]
    q = tmp_path / "quad.obj"
    q.write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf 1 2 3 4\n")
    with pytest.raises(ValueError, match="triangles"):
        scenes.load_obj(str(q))


def _stepped_child_sets(t0, t1):
    """The children proc_subtree's loop steps through (src/device_geometry_functions.cuh:159-188), by stepping
    find_first_node (:95-110) and next_node (:116-126) exactly as written there; float32 throughout."""
    n = t0.shape[0]
    tM = (np.float32(0.5) * (t0 + t1)).astype(np.float32)
    out = np.zeros(n, np.uint32)
    for k in range(n):
        a0, aM, a1 = t0[k], tM[k], t1[k]
        c = 0
        if a0[0] > a0[1] and a0[0] > a0[2]:
            c |= 2 if aM[1] < a0[0] else 0
            c |= 1 if aM[2] < a0[0] else 0
        elif a0[1] > a0[2]:
            c |= 4 if aM[0] < a0[1] else 0
            c |= 1 if aM[2] < a0[1] else 0
        else:
            c |= 4 if aM[0] < a0[2] else 0
            c |= 2 if aM[1] < a0[2] else 0
        s = 0
        while c < 8:
            s |= 1 << c
            ex = a1[0] if c & 4 else aM[0]
            ey = a1[1] if c & 2 else aM[1]
            ez = a1[2] if c & 1 else aM[2]
            nx = 8 if c & 4 else c | 4
            ny = 8 if c & 2 else c | 2
            nz = 8 if c & 1 else c | 1
            if ex < ey:
                c = nx if ex < ez else nz
            else:
                c = ny if ey < ez else nz
        out[k] = s
    return out


def test_child_set_table_equals_stepping_the_reference_loop(pt):
    """The node step reads the set of children to visit from a table keyed by ten comparisons; here the table route,
    evaluated in numpy with the same comparisons, against the stepped loop on tie-heavy intervals."""
    L = pt.lib()
    tab = np.zeros(1024, np.uint8)
    mir = np.zeros(2048, np.uint8)
    pt.check(L.pt_child_tables(tab.ctypes.data, mir.ctypes.data))
    for a in range(8):
        for m in (0, 1, 0x80, 0xA5, 0x3C, 0xFF, 0x12):
            assert mir[a << 8 | m] == sum(((m >> (c ^ a)) & 1) << c for c in range(8))
    rng = np.random.default_rng(7)
    n = 120000
    # small integer grids make ties between mid and exit parameters of different axes the common case; a slice of
    # random reals and degenerate (t0 == t1) axes cover the rest.  Mostly intervals a walk can hold (the ray's spans on
    # the three axes overlap: max t0 <= min t1), so that sets of two to four children are common
    t0 = rng.integers(-3, 4, size=(n, 3)).astype(np.float32)
    w = rng.integers(0, 9, size=(n, 3)).astype(np.float32)
    t0[: n // 4] = rng.normal(size=(n // 4, 3)).astype(np.float32)
    w[: n // 4] = (3.0 * np.abs(rng.normal(size=(n // 4, 3)))).astype(np.float32)
    t1 = (t0 + w).astype(np.float32)
    keep = (t0.max(axis=1) <= t1.min(axis=1)) | (rng.random(n) < 0.1)
    t0, t1 = t0[keep], t1[keep]
    n = t0.shape[0]
    assert n > 50000
    tM = (np.float32(0.5) * (t0 + t1)).astype(np.float32)
    te = t0.max(axis=1)
    bits = [tM[:, 0] < te, tM[:, 1] < te, tM[:, 2] < te, tM[:, 0] < np.minimum(t1[:, 1], t1[:, 2]), t1[:, 0] < tM[:, 1],
            tM[:, 1] < t1[:, 2], np.minimum(t1[:, 0], t1[:, 1]) < tM[:, 2], tM[:, 0] < tM[:, 1], tM[:, 0] < tM[:, 2],
            tM[:, 1] < tM[:, 2]]
    key = np.zeros(n, np.int64)
    for b in bits:
        key = 2 * key + b.astype(np.int64)
    stepped = _stepped_child_sets(t0, t1)
    assert np.array_equal(tab[key].astype(np.uint32), stepped)
    sizes = np.bincount([bin(int(x)).count("1") for x in stepped], minlength=5)
    assert sizes[2:].min() > 5000 and np.unique(key).size >= 60, (sizes, np.unique(key).size)


def test_scene_save_load_and_clone_carry_the_built_octrees(pt, scenes, teapot, tmp_path):
    """A scene built once can be handed to other GPUs of the node without rebuilding anything: as a clone inside one
    process, through a file between processes.  Same trees node for node, no octree build counted; damaged files
    are refused."""
    pos, tri = teapot
    sc = pt.Scene(-1)
    rng = np.random.default_rng(2)
    maps = dict(albedo=rng.integers(0, 256, (6, 10, 3), dtype=np.uint8), smoothness=rng.integers(0, 256, (3, 5), dtype=np.uint8))
    m0 = sc.add_material((0.1, 0.2, 0.3), (0.4, 0.5, 0.6), 2.0, 0.7, True, maps)
    m1 = sc.add_material()
    e0 = sc.add_mesh(pos, tri, m0, octree_depth=5)
    sc.add_sphere((1, 2, 3), 4.0, m1)
    fpos, ftri, _, _ = scenes.load_obj(os.path.join(REPO, "examples", "cornellbox", "floor.obj"))
    e2 = sc.add_mesh(fpos, ftri, m1, octree_depth=0)
    sc.set_hdri(rng.random((4, 8, 3), dtype=np.float32), 1.5, 1.2, 0.25, True)
    built = pt.octree_builds()
    path = str(tmp_path / "scene.bin")
    sc.save(path)
    loaded = pt.load_scene(path, device=-1, commit=False)
    clone = sc.clone(-1)
    assert pt.octree_builds() == built
    for other in (loaded, clone):
        assert np.array_equal(other.octree_dump(e0), sc.octree_dump(e0))
        assert other.octree_info(e0)["nodes_per_depth"].tolist() == sc.octree_info(e0)["nodes_per_depth"].tolist()
        with pytest.raises(pt.PtError, match="no octree"):
            other.octree_info(e2)
    # a saved scene saves again to the same bytes
    path2 = str(tmp_path / "again.bin")
    loaded.save(path2)
    data = open(path, "rb").read()
    assert data == open(path2, "rb").read()
    sc.close()          # the clone and the loaded scene outlive their source
    assert clone.octree_info(e0)["n_nodes"] > 0
    clone.close()
    loaded.close()
    # damaged files: truncated, wrong magic, cut inside the header
    bad = str(tmp_path / "bad.bin")
    for blob in (data[: len(data) // 2], b"NOTASCENE" + data[9:], data[:8]):
        open(bad, "wb").write(blob)
        with pytest.raises(pt.PtError, match="scene file"):
            pt.load_scene(bad, device=-1, commit=False)
    with pytest.raises(pt.PtError, match="cannot open"):
        pt.load_scene(str(tmp_path / "missing.bin"), device=-1, commit=False)
