"""GPU parity tests of the geometry path (call through the C ABI, compare with the CPU oracle and
with the committed golden vectors).  Bar: hit entity / triangle ids, t, barycentrics, triangle-test
order and work counters are BIT-EXACT; positions, normals and interpolated uvs are derived from
those by a handful of fp32 operations and are compared bit-exact too."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _assert_hits_equal(h_gpu, h_cpu):
    assert np.array_equal(h_gpu["entity"], h_cpu["entity"])
    assert np.array_equal(h_gpu["triangle"], h_cpu["triangle"])
    for f in ("t", "bu", "bv", "u", "v", "position", "normal"):
        assert np.array_equal(_bits(h_gpu[f]), _bits(h_cpu[f])), f


def _assert_unlogged_trace_equal(g, rays, h_cpu, c_cpu):
    """Without a test log the triangle runs are tested by the whole wave together (a different code
    path from the logged, lane-by-lane one): hits and work counters must be the same bits."""
    hq, lq, cq = g.trace(rays, log_stride=0)
    assert lq is None
    _assert_hits_equal(hq, h_cpu)
    for k in ("segments", "aabb_tests", "node_visits", "tri_tests"):
        assert cq[k] == c_cpu[k], k


def _random_rays(rng, n, center, radius):
    o = (center + rng.normal(size=(n, 3)) * radius).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    return np.concatenate([o, d], axis=1).astype(np.float32)


@pytest.fixture(scope="module")
def teapot_pair(pt, oracle, scenes):
    desc = scenes.teapot_scene(res=(96, 54))
    g, _ = pt.scene_from_description(desc)
    c, _ = oracle.scene_from_description(desc)
    return g, c, desc


def test_golden_vectors_teapot_and_cornell(pt, scenes):
    for name, builder, res in (("teapot_trace.npz", scenes.teapot_scene, (96, 54)), ("cornell_trace.npz", scenes.cornellbox_scene, (64, 36))):
        gold = np.load(os.path.join(GOLDEN, name))
        sc, _ = pt.scene_from_description(builder(res=res))
        for sfx in ("", "2"):
            if "rays" + sfx not in gold:
                continue
            stride = gold["log" + sfx].shape[1]
            hits, log, cnt = sc.trace(gold["rays" + sfx], log_stride=stride)
            assert np.array_equal(hits["entity"], gold["entity" + sfx])
            assert np.array_equal(hits["triangle"], gold["triangle" + sfx])
            for f in ("t", "bu", "bv"):
                assert np.array_equal(_bits(hits[f]), _bits(gold[f + sfx])), (name, f)
            assert np.array_equal(log, gold["log" + sfx]), "triangle-test order differs"
            want = dict(zip(pt.COUNTER_NAMES, gold["counters" + sfx].tolist()))
            for k in ("segments", "aabb_tests", "node_visits", "tri_tests"):
                assert cnt[k] == want[k], k
        sc.close()


def test_teapot_primary_and_random_rays_vs_oracle(teapot_pair, oracle):
    g, c, desc = teapot_pair
    cam = desc["camera"]
    ocam = oracle.make_camera(cam["position"], cam["direction"], (96, 54), cam["focal_length"])
    rng = np.random.default_rng(123)
    sets = [oracle.camera_rays(ocam, 1337, 5),
            _random_rays(rng, 20000, np.array([-205.3, 0.0, 82.5]), 60.0),      # origins inside and around the mesh
            _random_rays(rng, 5000, np.array([-205.3, 0.0, 82.5]), 400.0)]      # far origins, mostly misses
    # axis-parallel and negative-zero directions exercise the clamped reciprocals and the mirroring
    special = np.array([[-205.3, 0.0, -100.0, 0, 0, 1], [-205.3, 0.0, 300.0, 0, 0, -1], [-205.3, 200.0, 82.5, 0, -1, 0],
                        [-400.0, 0.0, 82.5, 1, 0, 0], [-205.3, 0.0, -100.0, -0.0, -0.0, 1], [-205.3, 10.0, -100.0, 0.0, -0.0, 1]],
                       dtype=np.float32)
    sets.append(special)
    for rays in sets:
        hg, lg, cg = g.trace(rays, log_stride=64)
        hc, lc, cc = c.trace(rays, log_stride=64)
        _assert_hits_equal(hg, hc)
        assert np.array_equal(lg, lc), "triangle-test order differs"
        for k in ("segments", "aabb_tests", "node_visits", "tri_tests"):
            assert cg[k] == cc[k], k
        _assert_unlogged_trace_equal(g, rays, hc, cc)
    assert (hg["entity"] >= 0).any()


def test_uv_mapped_smooth_mesh_and_depths(pt, oracle, scenes):
    """A uv-mapped, smooth-shaded procedural mesh at several octree depths, plus no octree."""
    pos, faces, uv = scenes.statue_mesh(6000)
    tri = scenes.tri6(faces, faces)
    rng = np.random.default_rng(5)
    c0 = pos.mean(axis=0)
    rays = _random_rays(rng, 8000, c0, float(np.abs(pos - c0).max()))
    for depth in (0, 3, 6, 9):
        g = pt.Scene(0)
        g.add_mesh(pos, tri, g.add_material(smooth_shading=True), uvs=uv, octree_depth=depth)
        g.add_sphere(c0 + np.array([0, 0, 30.0]), 4.0, g.add_material())
        g.commit()
        c = oracle.OracleScene()
        c.add_mesh(pos, tri, c.add_material(smooth_shading=True), uvs=uv, octree_depth=depth)
        c.add_sphere(c0 + np.array([0, 0, 30.0]), 4.0, c.add_material())
        hg, lg, cg = g.trace(rays, log_stride=40)
        hc, lc, cc = c.trace(rays, log_stride=40)
        _assert_hits_equal(hg, hc)
        if depth:   # brute-force logs overflow the stride; counts still match
            assert np.array_equal(lg, lc)
        assert np.array_equal(lg[:, 0], lc[:, 0])
        assert cg["tri_tests"] == cc["tri_tests"] and cg["node_visits"] == cc["node_visits"]
        assert (hg["entity"] == 1).any() and (hg["entity"] == 0).any()
        _assert_unlogged_trace_equal(g, rays, hc, cc)
        g.close()


def test_odd_runs_and_pair_tails(pt, oracle):
    """Triangle soups with 1, 3, 5, 7 and 64 triangles, without octree (one run over the whole mesh)
    and with shallow octrees (leaf runs of every parity): the last, half-filled pair of an odd run must
    never be tested, and the order of tests inside a pair is the order of the run."""
    rng = np.random.default_rng(77)
    for n_tris in (1, 3, 5, 7, 64):
        pos = (rng.random((3 * n_tris, 3)) * 10.0).astype(np.float32)
        tri = np.concatenate([np.arange(3 * n_tris, dtype=np.int32).reshape(-1, 3), np.full((n_tris, 3), -1, np.int32)], axis=1)
        rays = _random_rays(rng, 3000, np.array([5.0, 5.0, 5.0]), 6.0)
        for depth in (0, 1, 2):
            g = pt.Scene(0)
            g.add_mesh(pos, tri, g.add_material(), octree_depth=depth)
            g.commit()
            c = oracle.OracleScene()
            c.add_mesh(pos, tri, c.add_material(), octree_depth=depth)
            hg, lg, cg = g.trace(rays, log_stride=80)
            hc, lc, cc = c.trace(rays, log_stride=80)
            _assert_hits_equal(hg, hc)
            assert np.array_equal(lg, lc), (n_tris, depth)
            for k in ("segments", "aabb_tests", "node_visits", "tri_tests"):
                assert cg[k] == cc[k], (n_tris, depth, k)
            _assert_unlogged_trace_equal(g, rays, hc, cc)
            g.close()


def test_deepest_octree_and_many_entities(pt, oracle, scenes):
    """The depth limit (10 levels: the longest per-lane stack and the full width of the path word) next
    to thirty spheres and a few brute-force meshes, so that every lane works through a long entity list."""
    rng = np.random.default_rng(91)
    pos, faces, uv = scenes.statue_mesh(1500)
    tri = scenes.tri6(faces, faces)
    c0 = pos.mean(axis=0)
    ext = float(np.abs(pos - c0).max())
    g, c = pt.Scene(0), oracle.OracleScene()
    for s in (g, c):
        m = s.add_material()
        s.add_mesh(pos, tri, m, uvs=uv, octree_depth=10)
    centers = (c0 + rng.normal(size=(30, 3)) * ext).astype(np.float32)
    soups = [(c0 + rng.normal(size=(9, 3)) * ext).astype(np.float32) for _ in range(4)]
    soup_tri = np.concatenate([np.arange(9, dtype=np.int32).reshape(-1, 3), np.full((3, 3), -1, np.int32)], axis=1)
    for s in (g, c):
        m = s.add_material()
        for k in range(30):
            s.add_sphere(centers[k], 0.05 * ext * (1 + k % 4), m)
        for sp in soups:
            s.add_mesh(sp, soup_tri, m, octree_depth=0)
    g.commit()
    rays = _random_rays(rng, 6000, c0, 1.2 * ext)
    hg, lg, cg = g.trace(rays, log_stride=64)
    hc, lc, cc = c.trace(rays, log_stride=64)
    _assert_hits_equal(hg, hc)
    assert np.array_equal(lg, lc)
    for k in ("segments", "aabb_tests", "node_visits", "tri_tests"):
        assert cg[k] == cc[k], k
    assert len(np.unique(hg["entity"])) > 10
    _assert_unlogged_trace_equal(g, rays, hc, cc)
    info = g.octree_info(0)
    assert info["nodes_per_depth"][10] > 0
    g.close()


def test_empty_and_device_resident_buffers(teapot_pair, pt):
    g, c, desc = teapot_pair
    hits, _, cnt = g.trace(np.zeros((0, 6), np.float32))
    assert len(hits) == 0 and cnt["segments"] == 0
    # device-resident rays / hits (no staging inside the library)
    rng = np.random.default_rng(9)
    rays = _random_rays(rng, 4096, np.array([-205.3, 0.0, 82.5]), 80.0)
    d_rays = pt.DeviceBuffer(0, rays.nbytes)
    d_hits = pt.DeviceBuffer(0, len(rays) * pt.HIT_DTYPE.itemsize)
    d_rays.upload(rays)
    g.trace_device(d_rays.ptr, len(rays), d_hits.ptr)
    got = d_hits.download((len(rays),), dtype=pt.HIT_DTYPE)
    want, _, _ = c.trace(rays)
    _assert_hits_equal(got, want)


def test_camera_rays_match_oracle(pt, oracle):
    """Ray generator: pinhole rays are bit-exact; thin-lens rays use the hardware sin/cos (abs. error
    about 1e-6 on the lens offset), so origins may differ by 2 ulp at their magnitude (|x| < 64 ->
    8e-6) and unit directions by 4e-6."""
    for aperture in (0.0, 1.0):
        cam_g = pt.make_camera((40.5, 30.5, -50), (-0.4472136, 0.0, 0.8944272), (160, 90), 2.5, aperture, 85.0)
        cam_c = oracle.make_camera((40.5, 30.5, -50), (-0.4472136, 0.0, 0.8944272), (160, 90), 2.5, aperture, 85.0)
        for sample in (0, 17):
            rg = pt.camera_rays(cam_g, 1337, sample)
            rc = oracle.camera_rays(cam_c, 1337, sample)
            if aperture == 0.0:
                assert np.array_equal(_bits(rg), _bits(rc))
            else:
                assert np.abs(rg[:, :3] - rc[:, :3]).max() <= 8e-6
                assert np.abs(rg[:, 3:] - rc[:, 3:]).max() <= 4e-6


def test_missing_library_or_uncommitted_scene_fails_loudly(pt, teapot):
    pos, tri = teapot
    sc = pt.Scene(0)
    sc.add_mesh(pos, tri, sc.add_material(), octree_depth=4)
    with pytest.raises(pt.PtError, match="not committed"):
        sc.trace(np.zeros((1, 6), np.float32))
    sc.close()
    with pytest.raises(pt.PtError, match="no such device"):
        pt.Scene(64)
