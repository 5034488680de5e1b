"""ctypes binding of the CPU oracle (oracle/libptoracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg, never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libptoracle.so")


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.exists(os.path.join(_HERE, "pt_oracle.cpp"))
        and os.path.getmtime(os.path.join(_HERE, "pt_oracle.cpp")) > os.path.getmtime(_LIB_PATH)
    ):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libptoracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class Camera(C.Structure):
    _fields_ = [
        ("position", C.c_float * 3),
        ("direction", C.c_float * 3),
        ("up", C.c_float * 3),
        ("res_x", C.c_int),
        ("res_y", C.c_int),
        ("focal_length", C.c_float),
        ("aperture", C.c_float),
        ("focus_distance", C.c_float),
    ]


HIT_DTYPE = np.dtype(
    [
        ("entity", "<i4"),
        ("triangle", "<i4"),
        ("t", "<f4"),
        ("bu", "<f4"),
        ("bv", "<f4"),
        ("u", "<f4"),
        ("v", "<f4"),
        ("position", "<f4", (3,)),
        ("normal", "<f4", (3,)),
    ]
)

COUNTER_NAMES = ["segments", "aabb_tests", "node_visits", "tri_tests", "env_lookups", "tex_fetches", "paths"]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in COUNTER_NAMES]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n in COUNTER_NAMES}


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        L = _lib
        L.orc_scene_new.restype = C.c_void_p
        L.orc_scene_free.argtypes = [C.c_void_p]
        L.orc_add_material.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_float, C.c_int]
        L.orc_material_set_map.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.orc_add_mesh.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_add_sphere.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_float, C.c_int]
        L.orc_set_hdri.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int]
        L.orc_compute_vertex_normals.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_mesh_aabb.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_octree_stats.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_octree_dump.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]
        L.orc_octree_dump.restype = C.c_int64
        L.orc_aabb_intersects_triangle.argtypes = [C.c_void_p] * 5 + [C.c_int]
        L.orc_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_camera_rays.argtypes = [C.POINTER(Camera), C.c_uint32, C.c_int, C.c_void_p]
        L.orc_philox4x32_10.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_u01.argtypes = [C.c_uint32]
        L.orc_u01.restype = C.c_float
        L.orc_render.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_sample_hdri.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.orc_sample_material.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64] + [C.c_void_p] * 4
        L.orc_make_gaussian.argtypes = [C.c_uint, C.c_float, C.c_void_p]
        L.orc_make_mean.argtypes = [C.c_uint, C.c_void_p]
        L.orc_apply_filter.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_compound_buffers.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int]
        L.orc_normalize_image.argtypes = [C.c_void_p, C.c_int64, C.c_uint]
        L.orc_apply_threshold.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_float]
        L.orc_image_add.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_float]
        L.orc_apply_aces.argtypes = [C.c_void_p, C.c_int64]
        L.orc_gamma_correct.argtypes = [C.c_void_p, C.c_int64, C.c_float]
        L.orc_to_ppm_bytes.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    return _lib


def make_camera(position, direction, res, focal_length=1.0, aperture=0.0, focus_distance=100.0, up=(0, 1, 0)):
    cam = Camera()
    cam.position[:] = [float(x) for x in position]
    cam.direction[:] = [float(x) for x in direction]
    cam.up[:] = [float(x) for x in up]
    cam.res_x, cam.res_y = int(res[0]), int(res[1])
    cam.focal_length, cam.aperture, cam.focus_distance = float(focal_length), float(aperture), float(focus_distance)
    return cam


class OracleScene:
    """Builds the oracle's scene from the same plain-array description the product consumes."""

    def __init__(self):
        self.L = lib()
        self.h = C.c_void_p(self.L.orc_scene_new())
        self._keep = []

    def close(self):
        if self.h:
            self.L.orc_scene_free(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def add_material(self, albedo=(0.8, 0.8, 0.8), specular=(0, 0, 0), emission=0.0, smoothness=0.0, smooth_shading=False):
        a = (C.c_float * 3)(*[float(x) for x in albedo])
        s = (C.c_float * 3)(*[float(x) for x in specular])
        return self.L.orc_add_material(self.h, a, s, float(emission), float(smoothness), int(bool(smooth_shading)))

    def set_map(self, material, which, data):
        which_id = {"albedo": 0, "specular": 1, "normal": 2, "smoothness": 3}[which]
        data = np.ascontiguousarray(data, dtype=np.uint8)
        h, w = data.shape[0], data.shape[1]
        rc = self.L.orc_material_set_map(self.h, material, which_id, data.ctypes.data, w, h)
        assert rc == 0
        return rc

    def add_mesh(self, positions, tri, material, normals=None, uvs=None, octree_depth=0):
        positions = _f32(positions).reshape(-1, 3)
        tri = np.ascontiguousarray(tri, dtype=np.int32).reshape(-1, 6)
        nrm = _f32(normals).reshape(-1, 3) if normals is not None else None
        uv = _f32(uvs).reshape(-1, 2) if uvs is not None and len(uvs) else None
        return self.L.orc_add_mesh(
            self.h, positions.ctypes.data, nrm.ctypes.data if nrm is not None else None, len(positions),
            tri.ctypes.data, len(tri), uv.ctypes.data if uv is not None else None, len(uv) if uv is not None else 0,
            material, int(octree_depth))

    def add_sphere(self, center, radius, material):
        c = (C.c_float * 3)(*[float(x) for x in center])
        return self.L.orc_add_sphere(self.h, c, float(radius), material)

    def set_hdri(self, rgb, exposure=1.0, contrast=1.0, rot=0.0, smooth=False):
        rgb = _f32(rgb)
        h, w = rgb.shape[0], rgb.shape[1]
        self.L.orc_set_hdri(self.h, rgb.ctypes.data, w, h, float(exposure), float(contrast), float(rot), int(bool(smooth)))

    def octree_stats(self, entity, max_depth=6):
        per = np.zeros(max_depth + 1, dtype=np.int64)
        refs = C.c_int64(0)
        mx = C.c_int64(0)
        rc = self.L.orc_octree_stats(self.h, entity, per.ctypes.data, C.byref(refs), C.byref(mx))
        assert rc == 0
        return per, refs.value, mx.value

    def octree_dump(self, entity):
        n = self.L.orc_octree_dump(self.h, entity, None, 0)
        assert n >= 0
        out = np.zeros(n, dtype=np.int32)
        self.L.orc_octree_dump(self.h, entity, out.ctypes.data, n)
        return out

    def trace(self, rays, log_stride=0):
        rays = _f32(rays).reshape(-1, 6)
        hits = np.zeros(len(rays), dtype=HIT_DTYPE)
        log = np.zeros((len(rays), log_stride), dtype=np.int32) if log_stride else None
        cnt = Counters()
        self.L.orc_trace(self.h, rays.ctypes.data, len(rays), hits.ctypes.data,
                         log.ctypes.data if log is not None else None, log_stride, C.byref(cnt))
        return hits, log, cnt.as_dict()

    def render(self, cam, seed, sample_begin, sample_end, rows=None, n_threads=1, accum=None):
        if rows is None:
            rows = (0, cam.res_y)
        if accum is None:
            accum = np.zeros((cam.res_y, cam.res_x, 3), dtype=np.float32)
        cnt = Counters()
        self.L.orc_render(self.h, C.byref(cam), seed, sample_begin, sample_end, rows[0], rows[1],
                          accum.ctypes.data, n_threads, C.byref(cnt))
        return accum, cnt.as_dict()

    def sample_hdri(self, dirs):
        dirs = _f32(dirs).reshape(-1, 3)
        out = np.zeros_like(dirs)
        self.L.orc_sample_hdri(self.h, dirs.ctypes.data, len(dirs), out.ctypes.data)
        return out

    def sample_material(self, material, uv):
        uv = _f32(uv).reshape(-1, 2)
        n = len(uv)
        a = np.zeros((n, 3), np.float32)
        s = np.zeros((n, 3), np.float32)
        sm = np.zeros(n, np.float32)
        nn = np.zeros((n, 3), np.float32)
        self.L.orc_sample_material(self.h, material, uv.ctypes.data, n, a.ctypes.data, s.ctypes.data, sm.ctypes.data, nn.ctypes.data)
        return a, s, sm, nn


def scene_from_description(desc):
    """Feeds a plain-array scene description (the dict the product's scene builders return: materials,
    entities, hdri, camera) to the oracle; returns (OracleScene, Camera)."""
    sc = OracleScene()
    for m in desc["materials"]:
        mid = sc.add_material(m["albedo"], m["specular"], m["emission"], m["smoothness"], m["smooth_shading"])
        for which, data in m["maps"].items():
            sc.set_map(mid, which, data)
    for e in desc["entities"]:
        if e["kind"] == "mesh":
            sc.add_mesh(e["positions"], e["tri"], e["material"], normals=e["normals"], uvs=e["uvs"], octree_depth=e["octree_depth"])
        else:
            sc.add_sphere(e["center"], e["radius"], e["material"])
    if desc.get("hdri"):
        hd = desc["hdri"]
        sc.set_hdri(hd["rgb"], hd["exposure"], hd["contrast"], hd["rot"], hd["smooth"])
    c = desc["camera"]
    cam = make_camera(c["position"], c["direction"], c["res"], c["focal_length"], c["aperture"], c["focus_distance"])
    return sc, cam


def camera_rays(cam, seed, sample):
    out = np.zeros((cam.res_y * cam.res_x, 6), dtype=np.float32)
    lib().orc_camera_rays(C.byref(cam), seed, sample, out.ctypes.data)
    return out


def philox(ctr, key):
    ctr = np.ascontiguousarray(ctr, dtype=np.uint32)
    key = np.ascontiguousarray(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().orc_philox4x32_10(ctr.ctypes.data, key.ctypes.data, out.ctypes.data)
    return out


def compute_vertex_normals(positions, tri):
    positions = _f32(positions).reshape(-1, 3)
    tri = np.ascontiguousarray(tri, dtype=np.int32).reshape(-1, 6)
    out = np.zeros_like(positions)
    lib().orc_compute_vertex_normals(positions.ctypes.data, len(positions), tri.ctypes.data, len(tri), out.ctypes.data)
    return out


def transform(positions, normals, center, op, arg):
    """op: 'scale' | 'translate' | 'rotate'.  Returns (positions, normals, center) copies."""
    positions = _f32(positions).reshape(-1, 3).copy()
    normals = _f32(normals).reshape(-1, 3).copy() if normals is not None else None
    center = _f32(center).copy()
    a = _f32(np.broadcast_to(np.asarray(arg, dtype=np.float32), (3,))).copy()
    lib().orc_transform(positions.ctypes.data, normals.ctypes.data if normals is not None else None, len(positions),
                        center.ctypes.data, {"scale": 0, "translate": 1, "rotate": 2}[op], a.ctypes.data)
    return positions, normals, center


def mesh_aabb(positions):
    positions = _f32(positions).reshape(-1, 3)
    mn = np.zeros(3, np.float32)
    mx = np.zeros(3, np.float32)
    lib().orc_mesh_aabb(positions.ctypes.data, len(positions), mn.ctypes.data, mx.ctypes.data)
    return mn, mx


def aabb_intersects_triangle(bmin, bmax, v0, v1, v2, pad_box=False):
    a = [_f32(x) for x in (bmin, bmax, v0, v1, v2)]
    return bool(lib().orc_aabb_intersects_triangle(*[x.ctypes.data for x in a], int(pad_box)))


def make_gaussian(size, sigma):
    t = np.zeros((2 * size + 1, 2 * size + 1), np.float32)
    lib().orc_make_gaussian(size, float(sigma), t.ctypes.data)
    return t


def make_mean(size):
    t = np.zeros((2 * size + 1, 2 * size + 1), np.float32)
    lib().orc_make_mean(size, t.ctypes.data)
    return t


def apply_filter(img, taps, bilateral=False, n_threads=1):
    img = _f32(img)
    h, w = img.shape[0], img.shape[1]
    taps = _f32(taps)
    ks = (taps.shape[0] - 1) // 2
    out = np.zeros_like(img)
    lib().orc_apply_filter(img.ctypes.data, out.ctypes.data, w, h, taps.ctypes.data, ks, int(bool(bilateral)), n_threads)
    return out


def compound_buffers(split, n_split):
    split = _f32(split)
    n_pixels = split.size // 3 // n_split
    out = np.zeros((n_pixels, 3), np.float32)
    lib().orc_compound_buffers(split.ctypes.data, out.ctypes.data, n_pixels, n_split)
    return out


def normalize_image(img, n_samples):
    img = _f32(img).copy()
    lib().orc_normalize_image(img.ctypes.data, img.size // 3, n_samples)
    return img


def apply_threshold(img, threshold):
    img = _f32(img)
    out = np.zeros_like(img)
    lib().orc_apply_threshold(img.ctypes.data, out.ctypes.data, img.size // 3, float(threshold))
    return out


def image_add(img, layer, opacity=1.0):
    img = _f32(img).copy()
    layer = _f32(layer)
    lib().orc_image_add(img.ctypes.data, layer.ctypes.data, img.size // 3, float(opacity))
    return img


def apply_aces(img):
    img = _f32(img).copy()
    lib().orc_apply_aces(img.ctypes.data, img.size // 3)
    return img


def gamma_correct(img, gamma):
    img = _f32(img).copy()
    lib().orc_gamma_correct(img.ctypes.data, img.size // 3, float(gamma))
    return img


def to_ppm_bytes(img):
    img = _f32(img)
    out = np.zeros(img.shape, np.uint8)
    lib().orc_to_ppm_bytes(img.ctypes.data, img.size // 3, out.ctypes.data)
    return out


def post_chain(accum_sum, n_samples, res):
    """The fixed chain of Renderer::render (src/renderer.cu:277-297) on an un-normalised HxWx3 sum."""
    h, w = res[1], res[0]
    img = normalize_image(accum_sum.reshape(h, w, 3), n_samples)
    bright = apply_threshold(img, 3.0)
    bright = apply_filter(bright, make_gaussian(32, 10.0), False, n_threads=os.cpu_count() or 1)
    img = image_add(img, bright, 0.5)
    img = apply_aces(img)
    img = gamma_correct(img, 2.2)
    img = apply_filter(img, make_gaussian(2, 0.7), True, n_threads=os.cpu_count() or 1)
    return img
