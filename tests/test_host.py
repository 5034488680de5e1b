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
