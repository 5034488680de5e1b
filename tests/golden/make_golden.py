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


def main():
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
