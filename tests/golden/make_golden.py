"""Regenerates the oracle-derived golden vectors in this directory.

reference_known_answers.json is NOT produced here: it holds values obtained from the reference's
own code (see its _comment).  The .npz files below are produced by the CPU oracle AFTER it has been
checked against those known answers (tests/test_oracle.py), so that the GPU box -- which has neither
the reference nor, necessarily, a matching libm -- can compare the HIP path against fixed vectors
as well as against the oracle built there.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, "oracle"))
sys.path.insert(0, os.path.join(REPO, "cuda-path-tracer_amd", "python"))

import oracle_api as O  # noqa: E402
from ptamd import scenes as S  # noqa: E402


def golden_rays(cam_desc, res, seed, sample):
    cam = O.make_camera(cam_desc["position"], cam_desc["direction"], res, cam_desc["focal_length"], 0.0, cam_desc["focus_distance"])
    return O.camera_rays(cam, seed, sample)


def t0_vectors():
    """Small input/output vectors of single hot-path functions (produced by the oracle): the SAT box/triangle
    test, the environment lookup (nearest and bilinear, with the wrap column and the y_c = 0.99 clamp), texel
    indexing at the texture edges (u = 1, v = 0), the gaussian / mean tap tables, ACES and gamma."""
    rng = np.random.default_rng(20240607)
    out = {}
    # --- SAT (src/aabb.cu:64-120): random boxes and triangles around them, plus touching / sliver cases
    n = 600
    bmin = (rng.random((n, 3)) * 4 - 2).astype(np.float32)
    bmax = (bmin + rng.random((n, 3)).astype(np.float32) * 2 + np.float32(0.01)).astype(np.float32)
    ctr = 0.5 * (bmin + bmax)
    tri = (ctr[:, None, :] + rng.normal(size=(n, 3, 3)) * rng.choice([0.3, 1.5, 4.0], size=(n, 1, 1))).astype(np.float32)
    # triangles lying in a face plane of the box, on an edge, and degenerate ones
    tri[0] = [[bmax[0, 0], bmin[0, 1], bmin[0, 2]], [bmax[0, 0], bmax[0, 1], bmin[0, 2]], [bmax[0, 0], bmin[0, 1], bmax[0, 2]]]
    tri[1] = [[bmax[1, 0] + 1e-4, bmin[1, 1], bmin[1, 2]], [bmax[1, 0] + 1e-4, bmax[1, 1], bmin[1, 2]], [bmax[1, 0] + 1e-4, bmin[1, 1], bmax[1, 2]]]
    tri[2] = np.tile(ctr[2], (3, 1))
    tri[3] = np.tile(bmax[3] + np.float32(0.5), (3, 1))
    tri[4] = [ctr[4] - 10, ctr[4] + 10, ctr[4] + [10, -10, 0]]
    sat = np.zeros((n, 2), np.uint8)
    for i in range(n):
        for pad in (0, 1):
            sat[i, pad] = O.aabb_intersects_triangle(bmin[i], bmax[i], tri[i, 0], tri[i, 1], tri[i, 2], pad_box=bool(pad))
    out.update(sat_bmin=bmin, sat_bmax=bmax, sat_tri=tri, sat_result=sat)
    # --- sample_hdri (src/scene.cuh:36-82)
    W, H = 16, 8
    hdri = (rng.random((H, W, 3)) * 3 + 0.05).astype(np.float32)
    d = rng.normal(size=(4000, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    special = np.array([[0, 0, 1], [0, 0, -1], [1, 0, 0], [-1, 0, 0], [1e-3, -0.9999995, 1e-3], [1e-3, 0.9999995, 1e-3],
                        [-1e-6, 0.2, -1.0], [1e-6, 0.2, -1.0], [0.3, -1.0, 0.4], [0.6, 0.0, 0.8]], np.float64)
    special /= np.linalg.norm(special, axis=1, keepdims=True)
    configs = [dict(exposure=2.0, contrast=1.0, rot=0.0, smooth=False), dict(exposure=1.0, contrast=1.0, rot=0.85, smooth=True),
               dict(exposure=1.5, contrast=1.3, rot=0.4, smooth=True), dict(exposure=1.0, contrast=1.0, rot=0.4, smooth=False)]
    out["hdri_image"] = hdri
    out["hdri_configs"] = np.array([[c["exposure"], c["contrast"], c["rot"], float(c["smooth"])] for c in configs], np.float32)
    for k, cfg in enumerate(configs):
        dirs = np.concatenate([special, d], axis=0).astype(np.float32)
        if not cfg["smooth"]:
            # the nearest lookup is discontinuous at texel borders: keep directions whose texel coordinates lie
            # well inside a texel, so that a last-bit difference in acos / rsqrt cannot pick the neighbour
            dd = dirs.astype(np.float64)
            vp = dd[:, [0, 2]] / np.linalg.norm(dd[:, [0, 2]], axis=1, keepdims=True)
            xc = (np.arccos(np.clip(vp[:, 1], -1, 1)) * np.where(np.signbit(vp[:, 0]), -1, 1) / (2 * 3.14159265) + 0.5 + cfg["rot"]) % 1.0
            yc = np.minimum(1 - (dd[:, 1] + 1) / 2, 0.99)
            fx, fy = (xc * W) % 1.0, (yc * H) % 1.0
            keep = (fx > 0.02) & (fx < 0.98) & (fy > 0.02) & (fy < 0.98)
            dirs = dirs[keep]
        sc = O.OracleScene()
        sc.set_hdri(hdri, **cfg)
        out[f"hdri_dirs{k}"] = dirs
        out[f"hdri_rgb{k}"] = sc.sample_hdri(dirs)
    # --- texel indexing (src/texture.cuh:22-43) and material sampling (src/material.cuh:21-44)
    tw, th = 8, 4
    maps = dict(albedo=rng.integers(0, 256, (th, tw, 3), dtype=np.uint8), specular=rng.integers(0, 128, (th, tw, 3), dtype=np.uint8),
                normal=rng.integers(0, 256, (th, tw, 3), dtype=np.uint8), smoothness=rng.integers(0, 256, (2 * th, 2 * tw), dtype=np.uint8))
    us = np.concatenate([(np.arange(tw) + 0.5) / tw, [0.0, 1.0, 0.999999, 0.124, 0.126]])
    vs = np.concatenate([(np.arange(th) + 0.5) / th, [0.0, 1.0, 1e-7, 0.249, 0.251]])
    uv = np.stack(np.meshgrid(us, vs, indexing="ij"), axis=-1).reshape(-1, 2).astype(np.float32)
    sc = O.OracleScene()
    m_tex = sc.add_material((0.8, 0.8, 0.8), (0.2, 0.2, 0.2), 0.0, 0.5, False)
    for which, data in maps.items():
        sc.set_map(m_tex, which, data)
    m_plain = sc.add_material((0.3, 0.5, 0.7), (0.1, 0.2, 0.3), 0.0, 0.25, False)
    for name, mid in (("tex", m_tex), ("plain", m_plain)):
        a, s_, sm, nn = sc.sample_material(mid, uv)
        out.update({f"mat_{name}_albedo": a, f"mat_{name}_specular": s_, f"mat_{name}_smoothness": sm, f"mat_{name}_normal": nn})
    out.update(mat_uv=uv, **{f"map_{k}": v for k, v in maps.items()})
    # --- tap tables (src/post_processing.cu:40-84) and the tone-map operators (:194-238)
    out["gauss_2_0p7"] = O.make_gaussian(2, 0.7)
    out["gauss_32_10"] = O.make_gaussian(32, 10.0)
    out["gauss_1_3"] = O.make_gaussian(1, 3.0)
    out["mean_3"] = O.make_mean(3)
    x = np.concatenate([np.linspace(0, 12, 300), [0.0, 1e-6, 0.18, 1.0, 3.0, 100.0, 1e4]]).astype(np.float32)
    ramp = np.stack([x, x[::-1], np.roll(x, 7)], axis=-1).reshape(1, -1, 3).astype(np.float32)
    out["tone_in"] = ramp
    out["tone_aces"] = O.apply_aces(ramp)
    out["tone_gamma"] = O.gamma_correct(np.clip(ramp, 0, 1), 2.2)
    out["tone_threshold"] = O.apply_threshold(ramp, 3.0)
    out["tone_add"] = O.image_add(ramp, ramp[:, ::-1].copy(), 0.5)
    out["tone_normalize"] = O.normalize_image(ramp, 2048)
    out["tone_ppm"] = O.to_ppm_bytes(np.clip(ramp / 12, 0, 1.2))
    np.savez_compressed(os.path.join(HERE, "t0_vectors.npz"), **out)


def main():
    t0_vectors()
    # teapot (octree depth 6) + floor + light: primary rays of a 96x54 pinhole sweep, and one bounce generation
    desc = S.teapot_scene(res=(96, 54))
    sc, cam = O.scene_from_description(desc)
    rays = golden_rays(desc["camera"], (96, 54), 1337, 0)
    hits, log, cnt = sc.trace(rays, log_stride=48)
    # secondary rays: from every hit, a deterministic pseudo-random direction (covers negative directions / mirroring)
    rng = np.random.default_rng(7)
    sel = np.where(hits["entity"] >= 0)[0]
    d = rng.normal(size=(len(sel), 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    o = hits["position"][sel] + np.float32(0.01) * d
    rays2 = np.concatenate([o, d], axis=1).astype(np.float32)
    hits2, log2, cnt2 = sc.trace(rays2, log_stride=48)
    np.savez_compressed(os.path.join(HERE, "teapot_trace.npz"), rays=rays, entity=hits["entity"], triangle=hits["triangle"],
                        t=hits["t"], bu=hits["bu"], bv=hits["bv"], log=log, rays2=rays2, entity2=hits2["entity"],
                        triangle2=hits2["triangle"], t2=hits2["t"], bu2=hits2["bu"], bv2=hits2["bv"], log2=log2,
                        counters=np.array([cnt[k] for k in O.COUNTER_NAMES], dtype=np.int64),
                        counters2=np.array([cnt2[k] for k in O.COUNTER_NAMES], dtype=np.int64))
    # cornell box (brute force + spheres)
    desc = S.cornellbox_scene(res=(64, 36))
    sc, cam = O.scene_from_description(desc)
    rays = golden_rays(desc["camera"], (64, 36), 1337, 3)
    hits, log, cnt = sc.trace(rays, log_stride=16)
    np.savez_compressed(os.path.join(HERE, "cornell_trace.npz"), rays=rays, entity=hits["entity"], triangle=hits["triangle"],
                        t=hits["t"], bu=hits["bu"], bv=hits["bv"], log=log,
                        counters=np.array([cnt[k] for k in O.COUNTER_NAMES], dtype=np.int64))
    # octree structure hash of the teapot
    pos, tri, _, _ = S.load_obj(os.path.join(REPO, "examples", "teapot", "utah.obj"))
    s2 = O.OracleScene()
    m = s2.add_material()
    e = s2.add_mesh(pos, tri, m, octree_depth=6)
    dump = s2.octree_dump(e)
    import hashlib
    with open(os.path.join(HERE, "teapot_octree_sha256.txt"), "w") as f:
        f.write(hashlib.sha256(dump.tobytes()).hexdigest() + "\n")
    print("golden vectors written")


if __name__ == "__main__":
    main()
