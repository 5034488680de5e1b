"""GPU parity of single hot-path functions against the committed vectors (tests/golden/t0_vectors.npz, made
by tests/golden/make_golden.py) and against the oracle built on this machine: the environment lookup
(src/scene.cuh:36-82), material / texel sampling (src/material.cuh:21-44, src/texture.cuh:22-43) and the
tone-map operators (src/post_processing.cu:194-278), each through its C ABI entry point.

Tolerances: texel values are byte * (1/255) on the GPU and byte / 255 in the oracle (1 ulp: 6e-8); the
lookup's angle goes through the hardware rsq and the device acosf (x_c within 2e-7, i.e. a bilinear weight
within 1e-5 on a 16-texel row: 3e-5 of the texel range); pow(x, contrast) and gamma are the device powf."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def t0():
    return np.load(os.path.join(GOLDEN, "t0_vectors.npz"))


def test_sample_hdri_matches_vectors_and_oracle(pt, oracle, t0):
    for k, (exposure, contrast, rot, smooth) in enumerate(t0["hdri_configs"].tolist()):
        g = pt.Scene(0)
        g.set_hdri(t0["hdri_image"], exposure=exposure, contrast=contrast, rot=rot, smooth=bool(smooth))
        g.commit()
        dirs = t0[f"hdri_dirs{k}"]
        got = g.sample_hdri(dirs)
        scale = float(t0["hdri_image"].max()) * exposure * (3.0 if contrast != 1.0 else 1.0)
        tol = (1e-6 if not smooth else 1e-4) * scale
        assert np.abs(got - t0[f"hdri_rgb{k}"]).max() <= tol, k
        c = oracle.OracleScene()
        c.set_hdri(t0["hdri_image"], exposure=exposure, contrast=contrast, rot=rot, smooth=bool(smooth))
        assert np.abs(got - c.sample_hdri(dirs)).max() <= tol, k
        g.close()
    # a scene without environment map returns black
    g = pt.Scene(0)
    g.commit()
    assert not g.sample_hdri(np.array([[0, 0, 1]], np.float32)).any()
    g.close()


def test_sample_material_and_texel_edges(pt, oracle, t0):
    g = pt.Scene(0)
    maps = {k: t0["map_" + k] for k in ("albedo", "specular", "normal", "smoothness")}
    m_tex = g.add_material((0.8, 0.8, 0.8), (0.2, 0.2, 0.2), 0.0, 0.5, False, maps)
    m_plain = g.add_material((0.3, 0.5, 0.7), (0.1, 0.2, 0.3), 0.0, 0.25, False)
    g.commit()
    uv = t0["mat_uv"]
    for name, mid in (("tex", m_tex), ("plain", m_plain)):
        a, s, sm, nn = g.sample_material(mid, uv)
        assert np.abs(a - t0[f"mat_{name}_albedo"]).max() <= 6e-8
        assert np.abs(s - t0[f"mat_{name}_specular"]).max() <= 6e-8
        assert np.abs(sm - t0[f"mat_{name}_smoothness"]).max() <= 6e-8
        assert np.abs(nn - t0[f"mat_{name}_normal"]).max() <= 1e-6
    # which texel: u = 1 -> last column, v = 0 -> last row, v = 1 -> first row (indices clamped, never out of the map)
    a, _, sm, _ = g.sample_material(m_tex, np.array([[1.0, 0.0], [0.0, 1.0], [1.0, 1.0], [0.0, 0.0]], np.float32))
    alb = t0["map_albedo"].astype(np.float32) / np.float32(255)
    glo = t0["map_smoothness"].astype(np.float32) / np.float32(255)
    for row, (y, x) in enumerate(((-1, -1), (0, 0), (0, -1), (-1, 0))):
        assert np.abs(a[row] - alb[y, x]).max() <= 6e-8
        assert abs(sm[row] - glo[y, x]) <= 6e-8
    with pytest.raises(pt.PtError, match="unknown material"):
        g.sample_material(7, uv)
    g.close()


def test_texel_records_equal_separate_maps_and_oracle(pt, oracle, monkeypatch):
    """Maps of one resolution are uploaded as texel records (one 8- or 16-byte record per texel, 4x4-texel tiles);
    PTAMD_PLAIN_MAPS=1 keeps them as separate row-major maps.  Same texels either way: identical bits, and both
    within 1 ulp of the oracle (byte * (1 / 255) against byte / 255).  Sizes that are not multiples of the tile,
    every subset of maps, both record sizes."""
    rng = np.random.default_rng(11)
    cases = [((8, 4), ("albedo", "specular", "normal", "smoothness")), ((13, 7), ("albedo", "normal", "smoothness")),
             ((5, 9), ("specular",)), ((64, 32), ("normal",)), ((3, 3), ("smoothness",)), ((31, 17), ("albedo", "specular"))]
    for (tw, th), names in cases:
        maps = {k: rng.integers(0, 256, (th, tw) if k == "smoothness" else (th, tw, 3), dtype=np.uint8) for k in names}
        us = np.concatenate([(np.arange(tw) + 0.5) / tw, [0.0, 1.0, 0.999999, 1e-7]])
        vs = np.concatenate([(np.arange(th) + 0.5) / th, [0.0, 1.0, 1e-7, 0.999999]])
        uv = np.stack(np.meshgrid(us, vs, indexing="ij"), axis=-1).reshape(-1, 2).astype(np.float32)
        got = {}
        for plain in ("0", "1"):
            monkeypatch.setenv("PTAMD_PLAIN_MAPS", plain)
            g = pt.Scene(0)
            mid = g.add_material((0.8, 0.7, 0.6), (0.2, 0.1, 0.3), 0.0, 0.5, False, maps)
            g.commit()
            got[plain] = g.sample_material(mid, uv)
            tex_bytes = g.device_bytes()["textures"]
            g.close()
            if plain == "0":       # records: 16 bytes with a specular map, else 8, whole tiles
                assert tex_bytes == (-(-tw // 4)) * (-(-th // 4)) * 16 * (16 if "specular" in names else 8), (tw, th, names)
        for a, b in zip(got["0"], got["1"]):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (tw, th, names)
        c = oracle.OracleScene()
        om = c.add_material((0.8, 0.7, 0.6), (0.2, 0.1, 0.3), 0.0, 0.5, False)
        for k, v in maps.items():
            c.set_map(om, k, v)
        for a, b, tol in zip(got["0"], c.sample_material(om, uv), (6e-8, 6e-8, 6e-8, 1e-6)):
            assert np.abs(a - b).max() <= tol, (tw, th, names)


def test_environment_records_equal_row_major_image(pt, t0, monkeypatch):
    """Bilinear environment look-ups read one 64-byte record of four texels; PTAMD_PLAIN_MAPS=1 keeps the row-major
    image and four texel reads.  Identical bits, wrap column and last row included."""
    img = t0["hdri_image"]
    rng = np.random.default_rng(5)
    dirs = rng.normal(size=(20000, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    dirs = np.concatenate([dirs, np.array([[0, 1, 0], [0, -1, 0], [1e-9, 0.3, -1], [-1e-9, 0.3, -1], [0, -0.9999, 1]], np.float32)])
    got = {}
    for plain in ("0", "1"):
        monkeypatch.setenv("PTAMD_PLAIN_MAPS", plain)
        g = pt.Scene(0)
        g.set_hdri(img, exposure=1.5, contrast=1.0, rot=0.85, smooth=True)
        g.commit()
        got[plain] = g.sample_hdri(dirs)
        hdri_bytes = g.device_bytes()["hdri"]
        g.close()
        assert hdri_bytes == img.shape[0] * img.shape[1] * (64 if plain == "0" else 12)
    assert np.array_equal(got["0"].view(np.uint32), got["1"].view(np.uint32))


def _on_device(pt, img, op):
    buf = pt.DeviceBuffer(0, img.nbytes)
    buf.upload(img)
    op(buf)
    out = buf.download(img.shape)
    buf.free()
    return out


def test_tone_map_operators_match_vectors(pt, t0):
    L = pt.lib()
    ramp = np.ascontiguousarray(t0["tone_in"])
    n = ramp.size // 3
    aces = _on_device(pt, ramp, lambda b: pt.check(L.pt_apply_aces(0, b.ptr, n, None)))
    assert np.abs(aces - t0["tone_aces"]).max() <= 1e-6
    gam = _on_device(pt, np.clip(ramp, 0, 1), lambda b: pt.check(L.pt_gamma_correct(0, b.ptr, n, 2.2, None)))
    assert np.abs(gam - t0["tone_gamma"]).max() <= 1e-6
    nrm = _on_device(pt, ramp, lambda b: pt.check(L.pt_normalize_image(0, b.ptr, n, 2048, None)))
    np.testing.assert_allclose(nrm, t0["tone_normalize"], rtol=2e-7)
    src, dst = pt.DeviceBuffer(0, ramp.nbytes), pt.DeviceBuffer(0, ramp.nbytes)
    src.upload(ramp)
    pt.check(L.pt_apply_threshold(0, src.ptr, dst.ptr, n, 3.0, None))
    assert np.array_equal(dst.download(ramp.shape), t0["tone_threshold"])
    dst.upload(np.ascontiguousarray(ramp[:, ::-1]))
    pt.check(L.pt_image_add(0, src.ptr, dst.ptr, n, 0.5, None))
    assert np.array_equal(src.download(ramp.shape), t0["tone_add"])
