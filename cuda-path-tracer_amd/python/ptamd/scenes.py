"""Scene descriptions as plain arrays: OBJ reader, mesh transforms, procedural stand-ins for the
assets the reference does not ship, and the five benchmark configurations.

A scene description is a dict:
  materials: list of dict(albedo, specular, emission, smoothness, smooth_shading, maps={name: uint8 array})
  entities:  list of dict(kind='mesh', positions, normals|None, tri (n,6 int32), uvs|None, material, octree_depth)
                     or dict(kind='sphere', center, radius, material)
  hdri:      None or dict(rgb (H,W,3 float32), exposure, contrast, rot, smooth)
  camera:    dict(position, direction, res, focal_length, aperture, focus_distance)
The same description feeds the HIP library (through its C ABI) and the CPU oracle.
"""
import math
import os

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
EXAMPLES = os.path.join(REPO, "examples")
AABB_PADDING = np.float32(0.0001)


# ----------------------------------------------------------------------------- OBJ
def load_obj(path):
    """Minimal OBJ reader with tinyobjloader's index conventions (0-based, -1 when absent), as the
    reference's Entity constructor consumes them (src/entity.cu:13-148): one shape, triangles only.
    Returns positions (nv,3) f32, tri (nt,6) int32 [a,b,c,vta,vtb,vtc], uvs (nuv,2) f32, vn (nvn,3)."""
    pos, uvs, vns, tris = [], [], [], []
    with open(path, "r") as f:
        for line in f:
            line = line.split("#", 1)[0].strip()
            if not line:
                continue
            parts = line.split()
            tag = parts[0]
            if tag == "v":
                pos.append([float(parts[1]), float(parts[2]), float(parts[3])])
            elif tag == "vt":
                uvs.append([float(parts[1]), float(parts[2]) if len(parts) > 2 else 0.0])
            elif tag == "vn":
                vns.append([float(parts[1]), float(parts[2]), float(parts[3])])
            elif tag == "f":
                if len(parts) != 4:
                    raise ValueError("OBJ file faces must be triangles")
                vi, ti = [], []
                for p in parts[1:]:
                    fields = p.split("/")
                    v = int(fields[0])
                    vi.append(v - 1 if v > 0 else len(pos) + v)
                    if len(fields) > 1 and fields[1] != "":
                        t = int(fields[1])
                        ti.append(t - 1 if t > 0 else len(uvs) + t)
                    else:
                        ti.append(-1)
                tris.append(vi + ti)
    return (np.asarray(pos, dtype=np.float32).reshape(-1, 3), np.asarray(tris, dtype=np.int32).reshape(-1, 6),
            np.asarray(uvs, dtype=np.float32).reshape(-1, 2), np.asarray(vns, dtype=np.float32).reshape(-1, 3))


# ----------------------------------------------------------------------------- mesh helpers (fp32, reference op order)
def _normalize_rows(v):
    v = v.astype(np.float32)
    s = (v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1] + v[:, 2] * v[:, 2]).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        r = (np.float32(1.0) / np.sqrt(s, dtype=np.float32)).astype(np.float32)
    return (v * r[:, None]).astype(np.float32)


def mesh_center(positions):
    """Entity center = midpoint of the unpadded bounding box (src/entity.cu:139-143)."""
    mn = positions.min(axis=0).astype(np.float32)
    mx = positions.max(axis=0).astype(np.float32)
    return ((mn + mx) / np.float32(2)).astype(np.float32)


def scale_mesh(positions, center, factor):
    """src/entity.cu:185-202"""
    f = np.float32(factor)
    return (((positions - center) * f) + center).astype(np.float32)


def translate_mesh(positions, center, delta):
    """src/entity.cu:207-222"""
    d = np.asarray(delta, dtype=np.float32)
    return (positions + d).astype(np.float32), (center + d).astype(np.float32)


def rotation_rows(rot):
    """Rows R0,R1,R2 of src/entity.cu:232-242 (ZYX Euler; double cos/sin narrowed to fp32)."""
    ca, sa = np.float32(math.cos(float(np.float32(rot[2])))), np.float32(math.sin(float(np.float32(rot[2]))))
    cb, sb = np.float32(math.cos(float(np.float32(rot[1])))), np.float32(math.sin(float(np.float32(rot[1]))))
    cg, sg = np.float32(math.cos(float(np.float32(rot[0])))), np.float32(math.sin(float(np.float32(rot[0]))))
    R0 = np.array([ca * cb, ca * sb * sg - sa * cg, ca * sb * cg + sa * sg], dtype=np.float32)
    R1 = np.array([sa * cb, sa * sb * sg + ca * cg, sa * sb * cg - ca * sg], dtype=np.float32)
    R2 = np.array([-sb, cb * sg, cb * cg], dtype=np.float32)
    return R0, R1, R2


def _dot_rows(v, r):
    return (v[:, 0] * r[0] + v[:, 1] * r[1] + v[:, 2] * r[2]).astype(np.float32)


def rotate_mesh(positions, normals, center, rot):
    """src/entity.cu:227-258"""
    R0, R1, R2 = rotation_rows(rot)
    v = (positions - center).astype(np.float32)
    p = np.stack([_dot_rows(v, R0), _dot_rows(v, R1), _dot_rows(v, R2)], axis=1).astype(np.float32)
    p = (p + center).astype(np.float32)
    n = None
    if normals is not None:
        n = np.stack([_dot_rows(normals, R0), _dot_rows(normals, R1), _dot_rows(normals, R2)], axis=1).astype(np.float32)
    return p, n


def tri6(faces, vt=None):
    faces = np.asarray(faces, dtype=np.int32).reshape(-1, 3)
    if vt is None:
        vt = np.full_like(faces, -1)
    return np.ascontiguousarray(np.concatenate([faces, np.asarray(vt, dtype=np.int32).reshape(-1, 3)], axis=1))


# ----------------------------------------------------------------------------- procedural assets
def _hash01(ix, iy, seed):
    """Deterministic integer-lattice hash -> [0,1) (uint32 arithmetic, platform independent)."""
    m = np.uint64(0xFFFFFFFF)
    h = (ix.astype(np.uint64) * np.uint64(374761393) + iy.astype(np.uint64) * np.uint64(668265263)
         + np.uint64((int(seed) * 2246822519) & 0xFFFFFFFF)) & m
    h = ((h ^ (h >> np.uint64(13))) * np.uint64(1274126177)) & m
    h = h ^ (h >> np.uint64(16))
    return (h >> np.uint64(8)).astype(np.float64) / float(1 << 24)


def value_noise(u, v, freq, seed, periodic_u=True):
    """Smooth value noise on [0,1]^2, periodic in u (so tube / equirect seams close)."""
    x = u * freq
    y = v * freq
    ix = np.floor(x).astype(np.int64)
    iy = np.floor(y).astype(np.int64)
    fx = x - ix
    fy = y - iy
    sx = fx * fx * (3 - 2 * fx)
    sy = fy * fy * (3 - 2 * fy)
    if periodic_u:
        ix0, ix1 = ix % freq, (ix + 1) % freq
    else:
        ix0, ix1 = ix, ix + 1
    a = _hash01(ix0, iy, seed)
    b = _hash01(ix1, iy, seed)
    c = _hash01(ix0, iy + 1, seed)
    d = _hash01(ix1, iy + 1, seed)
    return (a * (1 - sx) + b * sx) * (1 - sy) + (c * (1 - sx) + d * sx) * sy


def fbm(u, v, octaves, seed, base=4):
    out = np.zeros_like(u, dtype=np.float64)
    amp, tot = 1.0, 0.0
    for o in range(octaves):
        out += amp * value_noise(u, v, base << o, seed + 17 * o)
        tot += amp
        amp *= 0.5
    return out / tot


def tube_mesh(nu, nv, curve, radius_fn, seed, displacement):
    """UV-mapped closed tube swept along a closed curve: nu segments along, nv around.
    Vertices are duplicated along both seams so every vertex has one uv (2*nu*nv triangles)."""
    s = np.linspace(0.0, 1.0, nu + 1)
    t = np.linspace(0.0, 1.0, nv + 1)
    S, T = np.meshgrid(s, t, indexing="ij")
    P = curve(S)                      # (nu+1, nv+1, 3)
    eps = 1e-4
    tangent = curve(S + eps) - curve(S - eps)
    tangent /= np.linalg.norm(tangent, axis=-1, keepdims=True)
    up = np.array([0.0, 0.0, 1.0])
    n1 = np.cross(tangent, up)
    n1 /= np.linalg.norm(n1, axis=-1, keepdims=True)
    n2 = np.cross(tangent, n1)
    ang = 2 * np.pi * T
    noise = fbm(S % 1.0, T % 1.0, 5, seed, base=8)
    # periodic in T as well: blend the seam
    noise_wrap = fbm(S % 1.0, (T + 0.5) % 1.0, 5, seed, base=8)
    wgt = np.abs(2 * T - 1.0)
    noise = noise * (1 - wgt) + noise_wrap * wgt
    r = radius_fn(S) * (1.0 + displacement * (noise - 0.5) * 2.0)
    pos = P + r[..., None] * (np.cos(ang)[..., None] * n1 + np.sin(ang)[..., None] * n2)
    pos = pos.reshape(-1, 3).astype(np.float32)
    uv = np.stack([S, T], axis=-1).reshape(-1, 2).astype(np.float32)
    idx = np.arange((nu + 1) * (nv + 1)).reshape(nu + 1, nv + 1)
    a = idx[:-1, :-1].ravel()
    b = idx[1:, :-1].ravel()
    c = idx[1:, 1:].ravel()
    d = idx[:-1, 1:].ravel()
    # counter-clockwise seen from outside: cross(v1 - v0, v2 - v0) points away from the curve.  The
    # renderer never flips normals (src/device_geometry_functions.cuh:319-365), so an inward
    # winding would send every bounce into the tube.
    faces = np.concatenate([np.stack([a, c, b], axis=1), np.stack([a, d, c], axis=1)], axis=0).astype(np.int32)
    return pos, faces, uv


def torus_knot_curve(p, q, R, r):
    def curve(s):
        phi = 2 * np.pi * s
        rad = R + r * np.cos(q * phi)
        return np.stack([rad * np.cos(p * phi), rad * np.sin(p * phi), -r * np.sin(q * phi)], axis=-1)
    return curve


def statue_mesh(n_tris_target, seed=7, height=45.0):
    """Stand-in for the reference's missing 80k-triangle statue (examples/hcandersen/source/80k.obj):
    a noise-displaced torus-knot of similar extent, UV-mapped, ~n_tris_target triangles.  Built in
    the orientation the reference's scene script expects BEFORE its rotate(-pi/2, pi/2, 0)."""
    nv = max(8, int(round(math.sqrt(n_tris_target / 2 / 8))))
    nu = max(8, int(round(n_tris_target / 2 / nv)))
    pos, faces, uv = tube_mesh(nu, nv, torus_knot_curve(2, 3, 1.0, 0.42), lambda s: 0.22 + 0.0 * s, seed, 0.18)
    ext = pos.max(axis=0) - pos.min(axis=0)
    pos = (pos * np.float32(height / ext.max())).astype(np.float32)
    return pos, faces, uv


def grid_terrain(n, size, amp, seed):
    """Displaced (n x n)-quad grid in the xz-plane: 2*n*n triangles (the synthetic high-poly scene)."""
    s = np.linspace(0.0, 1.0, n + 1)
    S, T = np.meshgrid(s, s, indexing="ij")
    hgt = amp * (fbm(S * 0.999, T * 0.999, 6, seed, base=4) - 0.5)
    pos = np.stack([(S - 0.5) * size, hgt, (T - 0.5) * size], axis=-1).reshape(-1, 3).astype(np.float32)
    uv = np.stack([S, T], axis=-1).reshape(-1, 2).astype(np.float32)
    idx = np.arange((n + 1) * (n + 1)).reshape(n + 1, n + 1)
    a = idx[:-1, :-1].ravel()
    b = idx[1:, :-1].ravel()
    c = idx[1:, 1:].ravel()
    d = idx[:-1, 1:].ravel()
    faces = np.concatenate([np.stack([a, c, b], axis=1), np.stack([a, d, c], axis=1)], axis=0).astype(np.int32)
    return pos, faces, uv


def _upsample(a, factor):
    """Bilinear upsampling of a 2-D (or HxWxC) array by an integer factor (periodic borders)."""
    if factor == 1:
        return a
    h, w = a.shape[0], a.shape[1]
    ys = (np.arange(h * factor) + 0.5) / factor - 0.5
    xs = (np.arange(w * factor) + 0.5) / factor - 0.5
    y0 = np.floor(ys).astype(np.int64)
    x0 = np.floor(xs).astype(np.int64)
    fy = (ys - y0).astype(np.float32)
    fx = (xs - x0).astype(np.float32)
    y0m, y1m = y0 % h, (y0 + 1) % h
    x0m, x1m = x0 % w, (x0 + 1) % w
    a = a.astype(np.float32)
    shape_y = (-1, 1) + (1,) * (a.ndim - 2)
    rows = a[y0m] * (1 - fy).reshape(shape_y) + a[y1m] * fy.reshape(shape_y)
    shape_x = (1, -1) + (1,) * (a.ndim - 2)
    return rows[:, x0m] * (1 - fx).reshape(shape_x) + rows[:, x1m] * fx.reshape(shape_x)


def procedural_textures(res, seed=3):
    """Albedo / normal (RGB8) and gloss (G8) maps standing in for the reference's missing
    examples/hcandersen/textures/{diff,normal,gloss}.jpg.  The noise is evaluated on a grid of at
    most 512^2 and upsampled, so building 4096^2 maps takes seconds."""
    base = min(res, 512)
    factor = max(1, res // base)
    res = base * factor
    s = (np.arange(base) + 0.5) / base
    U, Vv = np.meshgrid(s, s, indexing="xy")
    n1 = fbm(U, Vv, 5, seed, base=8)
    n2 = fbm(U, Vv, 4, seed + 101, base=16)
    hgt = fbm(U, Vv, 6, seed + 202, base=8)
    n1, n2, hgt = _upsample(n1, factor), _upsample(n2, factor), _upsample(hgt, factor)
    albedo = np.stack([0.55 + 0.35 * n1, 0.45 + 0.35 * n2, 0.35 + 0.3 * n1 * n2], axis=-1)
    albedo = np.clip(albedo * 255.0, 0, 255).astype(np.uint8)
    dx = np.roll(hgt, -factor, axis=1) - np.roll(hgt, factor, axis=1)
    dy = np.roll(hgt, -factor, axis=0) - np.roll(hgt, factor, axis=0)
    nrm = np.stack([-dx * base * 0.02, -dy * base * 0.02, np.ones_like(dx)], axis=-1)
    nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    normal = np.clip((nrm * 0.5 + 0.5) * 255.0, 0, 255).astype(np.uint8)
    gloss = np.clip((0.2 + 0.6 * n2) * 255.0, 0, 255).astype(np.uint8)
    return np.ascontiguousarray(albedo), np.ascontiguousarray(normal), np.ascontiguousarray(gloss)


def procedural_hdri(w=4096, h=2048, seed=11):
    """fp32 equirectangular sky standing in for the reference's missing examples/hdris/*.hdr:
    gradient sky + warm ground + a few high-intensity lobes (so the bloom threshold 3.0 triggers)."""
    x = ((np.arange(w) + 0.5) / w).astype(np.float32)
    y = ((np.arange(h) + 0.5) / h).astype(np.float32)
    X, Y = np.meshgrid(x, y, indexing="xy")
    sky = np.stack([0.35 + 0.3 * Y, 0.5 + 0.25 * Y, 0.9 - 0.2 * Y], axis=-1)
    ground = np.stack([0.25 + 0 * Y, 0.22 + 0 * Y, 0.2 + 0 * Y], axis=-1)
    k = np.clip((Y - 0.55) / 0.1, 0, 1)[..., None]
    img = sky * (1 - k) + ground * k
    factor = max(1, w // 512)
    bw, bh = w // factor, h // factor
    bx = (np.arange(bw) + 0.5) / bw
    by = (np.arange(bh) + 0.5) / bh
    BX, BY = np.meshgrid(bx, by, indexing="xy")
    clouds = _upsample(fbm(BX, BY * 0.5, 5, seed, base=8), factor)[:h, :w, None]
    img = img * (0.8 + 0.5 * clouds)
    for (cx, cy, rad, inten, col) in [(0.30, 0.22, 0.012, 60.0, (1.0, 0.93, 0.8)), (0.72, 0.30, 0.03, 9.0, (1.0, 0.85, 0.7)),
                                      (0.55, 0.12, 0.02, 14.0, (0.8, 0.9, 1.0))]:
        dxw = np.minimum(np.abs(X - cx), 1 - np.abs(X - cx))
        d2 = (dxw * 2) ** 2 + (Y - cy) ** 2
        img += (inten * np.exp(-d2 / np.float32(2 * rad * rad)))[..., None] * np.asarray(col, dtype=np.float32)
    return np.ascontiguousarray(img.astype(np.float32))


# ----------------------------------------------------------------------------- scene builders
def _mat(albedo=(0.8, 0.8, 0.8), specular=(0, 0, 0), emission=0.0, smoothness=0.0, smooth_shading=False, maps=None):
    return dict(albedo=tuple(float(x) for x in albedo), specular=tuple(float(x) for x in specular),
                emission=float(emission), smoothness=float(smoothness), smooth_shading=bool(smooth_shading),
                maps=dict(maps or {}))


def _mesh(positions, tri, material, normals=None, uvs=None, octree_depth=0):
    return dict(kind="mesh", positions=np.ascontiguousarray(positions, dtype=np.float32),
                normals=None if normals is None else np.ascontiguousarray(normals, dtype=np.float32),
                tri=np.ascontiguousarray(tri, dtype=np.int32), uvs=None if uvs is None or len(uvs) == 0 else
                np.ascontiguousarray(uvs, dtype=np.float32), material=material, octree_depth=int(octree_depth))


def _sphere(center, radius, material):
    return dict(kind="sphere", center=tuple(float(x) for x in center), radius=float(radius), material=material)


def _camera(position, direction, res, focal_length=1.0, aperture=0.0, focus_distance=100.0):
    d = np.asarray(direction, dtype=np.float32)
    s = np.float32(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])
    d = (d * (np.float32(1.0) / np.sqrt(s, dtype=np.float32))).astype(np.float32)
    return dict(position=tuple(float(x) for x in position), direction=tuple(float(x) for x in d), res=(int(res[0]), int(res[1])),
                focal_length=float(focal_length), aperture=float(aperture), focus_distance=float(focus_distance))


def _quad_entity(name, material, translate=None, scale=None):
    pos, tri, uvs, _ = load_obj(os.path.join(EXAMPLES, "cornellbox", name))
    center = mesh_center(pos)
    if scale is not None:
        pos = scale_mesh(pos, center, scale)
    if translate is not None:
        pos, center = translate_mesh(pos, center, translate)
    return _mesh(pos, tri, material)


def cornellbox_scene(res=(640, 360)):
    """Config 1: the reference ships the three quads (examples/cornellbox/*.obj) and material hints
    (src/main.cu:9-13,22-23) but no scene script; this one closes the box with translated copies,
    adds the sphere light and a glossy sphere.  Brute-force meshes (no octree)."""
    mats = [_mat((1, 1, 1)), _mat((0.8, 0.1, 0.1)), _mat((0.1, 0.8, 0.1)), _mat((1, 1, 1), emission=10.0),
            _mat((0.8, 0.8, 0.8), specular=(0.5, 0.5, 0.5), smoothness=0.6)]
    ents = [
        _quad_entity("floor.obj", 0),
        _quad_entity("floor.obj", 0, translate=(0, 55, 0)),
        _quad_entity("back_wall.obj", 0),
        _quad_entity("side_wall.obj", 1),
        _quad_entity("side_wall.obj", 2, translate=(55, 0, 0)),
        _sphere((27.5, 50.0, 28.5), 6.0, 3),
        _sphere((18.0, 9.0, 35.0), 9.0, 4),
        _sphere((38.0, 7.0, 22.0), 7.0, 0),
    ]
    cam = _camera((27.5, 27.5, -60.0), (0, 0, 1), res, focal_length=1.4)
    return dict(name="cornellbox", materials=mats, entities=ents, hdri=None, camera=cam)


def teapot_scene(res=(1280, 720), octree_depth=6):
    """Config 2: examples/teapot/utah.obj with the depth-6 octree on a scaled floor quad, sphere
    light, no HDRI, pinhole camera."""
    pos, tri, uvs, _ = load_obj(os.path.join(EXAMPLES, "teapot", "utah.obj"))
    mats = [_mat((0.8, 0.8, 0.8)), _mat((0.8, 0.6, 0.3), specular=(0.3, 0.3, 0.3), smoothness=0.5, smooth_shading=True),
            _mat((1, 1, 1), emission=10.0)]
    fpos, ftri, _, _ = load_obj(os.path.join(EXAMPLES, "cornellbox", "floor.obj"))
    fc = mesh_center(fpos)
    fpos = scale_mesh(fpos, fc, 20.0)
    fpos, fc = translate_mesh(fpos, fc, (-205.3 - 27.5, -36.4, 82.5 - 28.5))
    ents = [_mesh(pos, tri, 1, octree_depth=octree_depth), _mesh(fpos, ftri, 0),
            _sphere((-150.0, 160.0, -40.0), 40.0, 2)]
    cam = _camera((-205.3, 40.0, -150.0), (0, -0.12, 1), res, focal_length=1.6)
    return dict(name="teapot", materials=mats, entities=ents, hdri=None, camera=cam)


def _floor_under(y, half, material):
    pos = np.array([[-half, y, half], [half, y, half], [-half, y, -half], [half, y, -half]], dtype=np.float32)
    faces = np.array([[0, 3, 2], [0, 1, 3]], dtype=np.int32)
    return _mesh(pos, tri6(faces), material)


def dragon_scene(res=(1920, 1080), n_tris=871414, octree_depth=7, hdri_res=(2048, 1024)):
    """Config 3: the reference's examples/dragon/dragon.obj is not shipped; stand-in is a
    noise-displaced (3,5) torus knot with about as many triangles as the Stanford dragon."""
    nv = 256
    nu = max(8, int(round(n_tris / 2 / nv)))
    pos, faces, uv = tube_mesh(nu, nv, torus_knot_curve(3, 5, 1.0, 0.38), lambda s: 0.17 + 0.0 * s, 23, 0.25)
    pos = (pos * np.float32(30.0)).astype(np.float32)
    center = mesh_center(pos)
    pos, n = rotate_mesh(pos, None, center, (-math.pi / 2, 0.0, 0.0))
    pos, center = translate_mesh(pos, center, (0.0, 22.0 - float(center[1]), 0.0))
    mats = [_mat((0.3, 0.75, 0.45), specular=(0.25, 0.25, 0.25), smoothness=0.7, smooth_shading=True), _mat((0.7, 0.7, 0.7))]
    ents = [_mesh(pos, tri6(faces, faces), 0, uvs=uv, octree_depth=octree_depth), _floor_under(float(pos[:, 1].min()), 400.0, 1)]
    hdri = dict(rgb=procedural_hdri(hdri_res[0], hdri_res[1]), exposure=1.0, contrast=1.0, rot=0.2, smooth=True)
    cam = _camera((45.0, 38.0, -85.0), (-0.45, -0.15, 1), res, focal_length=2.0, aperture=0.6, focus_distance=95.0)
    return dict(name="dragon", materials=mats, entities=ents, hdri=hdri, camera=cam, n_tris=int(len(faces)))


def hcandersen_scene(res=(1920, 1080), n_tris=80000, octree_depth=6, tex_res=4096, hdri_res=(4096, 2048)):
    """Config 4 (north star): the scene script of src/main.cu:14-65 with procedural stand-ins for
    the assets the reference does not ship (80k.obj, diff/gloss/normal.jpg, the 4k HDRI): default
    albedo 0.8 + specular 0.2 + albedo/normal/gloss maps, mesh rotated (-pi/2, pi/2, 0) and moved
    up 22.65, octree, smoothed HDRI rotated 0.85, thin-lens camera (focal 2.5, aperture 1, focus 85)."""
    pos, faces, uv = statue_mesh(n_tris)
    center = mesh_center(pos)
    pos, _ = rotate_mesh(pos, None, center, (-3.14159265 / 2.0, 3.14159265 / 2.0, 0.0))
    pos, center = translate_mesh(pos, center, (0.0, 22.65, 0.0))
    albedo, normal, gloss = procedural_textures(tex_res)
    mats = [_mat((0.8, 0.8, 0.8), specular=(0.2, 0.2, 0.2), smoothness=1.0,
                 maps=dict(albedo=albedo, normal=normal, smoothness=gloss))]
    ents = [_mesh(pos, tri6(faces, faces), 0, uvs=uv, octree_depth=octree_depth)]
    hdri = dict(rgb=procedural_hdri(hdri_res[0], hdri_res[1]), exposure=1.0, contrast=1.0, rot=0.85, smooth=True)
    cam = _camera((40.5, 30.5, -50.0), (-0.5, 0.0, 1.0), res, focal_length=2.5, aperture=1.0, focus_distance=85.0)
    return dict(name="hcandersen", materials=mats, entities=ents, hdri=hdri, camera=cam, n_tris=int(len(faces)))


def synthetic_scene(res=(3840, 2160), n_tris=20_000_000, octree_depth=9, hdri_res=(4096, 2048)):
    """Config 5: displaced-grid terrain with ~n_tris triangles (octree depth raised; stated in reports)."""
    n = int(round(math.sqrt(n_tris / 2)))
    pos, faces, uv = grid_terrain(n, 400.0, 60.0, 31)
    mats = [_mat((0.75, 0.7, 0.6), specular=(0.1, 0.1, 0.1), smoothness=0.3, smooth_shading=True)]
    ents = [_mesh(pos, tri6(faces, faces), 0, uvs=uv, octree_depth=octree_depth)]
    hdri = dict(rgb=procedural_hdri(hdri_res[0], hdri_res[1]), exposure=1.0, contrast=1.0, rot=0.1, smooth=True)
    cam = _camera((0.0, 90.0, -260.0), (0.0, -0.35, 1.0), res, focal_length=1.8, aperture=0.5, focus_distance=250.0)
    return dict(name="synthetic", materials=mats, entities=ents, hdri=hdri, camera=cam, n_tris=int(len(faces)))


CONFIGS = {
    "cornellbox": dict(builder=cornellbox_scene, res=(640, 360), spp=32),
    "teapot": dict(builder=teapot_scene, res=(1280, 720), spp=256),
    "dragon": dict(builder=dragon_scene, res=(1920, 1080), spp=1024),
    "hcandersen": dict(builder=hcandersen_scene, res=(1920, 1080), spp=2048),
    "synthetic": dict(builder=synthetic_scene, res=(3840, 2160), spp=4096),
}

