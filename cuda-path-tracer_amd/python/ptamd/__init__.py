"""ctypes binding of libptamd.so (the C ABI in include/ptamd.h).

Used by bench.py, the tests and the multi-GPU driver.  PyTorch, where used, only supplies device
memory, streams and torch.distributed; every compute call goes through the C ABI.  There is no
CPU fallback: if the library is missing or a call fails, an exception is raised.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(os.path.dirname(_HERE))
LIB_PATH = os.environ.get("PTAMD_LIB") or os.path.join(PKG_ROOT, "lib", "libptamd.so")
PT_MAX_OCTREE_DEPTH = 10


class PtError(RuntimeError):
    pass


class Camera(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("direction", C.c_float * 3), ("up", C.c_float * 3),
                ("res_x", C.c_int), ("res_y", C.c_int), ("focal_length", C.c_float), ("aperture", C.c_float),
                ("focus_distance", C.c_float)]


class MaterialDesc(C.Structure):
    _fields_ = [("albedo", C.c_float * 3), ("specular", C.c_float * 3), ("emission", C.c_float), ("smoothness", C.c_float),
                ("smooth_shading", C.c_int),
                ("albedo_map", C.c_void_p), ("albedo_w", C.c_int), ("albedo_h", C.c_int),
                ("specular_map", C.c_void_p), ("specular_w", C.c_int), ("specular_h", C.c_int),
                ("normal_map", C.c_void_p), ("normal_w", C.c_int), ("normal_h", C.c_int),
                ("smoothness_map", C.c_void_p), ("smoothness_w", C.c_int), ("smoothness_h", C.c_int)]


COUNTER_NAMES = ["segments", "aabb_tests", "node_visits", "tri_tests", "env_lookups", "tex_fetches", "paths"]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in COUNTER_NAMES]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n in COUNTER_NAMES}


class OctreeInfo(C.Structure):
    _fields_ = [("max_depth", C.c_int32), ("n_nodes", C.c_int64), ("n_leaf_refs", C.c_int64), ("max_leaf", C.c_int64),
                ("nodes_per_depth", C.c_int64 * (PT_MAX_OCTREE_DEPTH + 1))]


HIT_DTYPE = np.dtype([("entity", "<i4"), ("triangle", "<i4"), ("t", "<f4"), ("bu", "<f4"), ("bv", "<f4"), ("u", "<f4"),
                      ("v", "<f4"), ("position", "<f4", (3,)), ("normal", "<f4", (3,))])

# every symbol include/ptamd.h declares
ABI_SYMBOLS = [
    "pt_last_error", "pt_device_count", "pt_device_name", "pt_scene_create", "pt_scene_destroy", "pt_scene_add_material",
    "pt_scene_add_mesh", "pt_scene_add_sphere", "pt_scene_set_hdri", "pt_scene_commit", "pt_scene_clone", "pt_scene_save", "pt_scene_load", "pt_octree_builds", "pt_scene_octree_info",
    "pt_scene_octree_dump", "pt_scene_device_bytes", "pt_trace_rays", "pt_camera_rays", "pt_sample_hdri", "pt_sample_material", "pt_box_intersects_triangle", "pt_child_tables", "pt_render_accumulate", "pt_render_accumulate_tiles", "pt_render_slices",
    "pt_render_accumulate_noalloc", "pt_noalloc_max_partial_bytes", "pt_scene_set_launch", "pt_scene_get_launch", "pt_scene_render_residency",
    "pt_scene_set_profiling", "pt_scene_schedule_counters", "pt_scene_wave_profile", "pt_scene_last_render_kernel_ms", "pt_compound_buffers", "pt_reduce_sum", "pt_apply_filter", "pt_make_gaussian", "pt_make_mean", "pt_normalize_image", "pt_apply_threshold",
    "pt_image_add", "pt_apply_aces", "pt_gamma_correct", "pt_post_chain", "pt_malloc", "pt_free", "pt_memset", "pt_upload",
    "pt_download", "pt_stream_synchronize",
]

_lib = None


def lib():
    """Loads libptamd.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PtError(f"{LIB_PATH} not found: build it with __graft_entry__.build() or make -C cuda-path-tracer_amd/csrc")
        L = C.CDLL(LIB_PATH)
        L.pt_last_error.restype = C.c_char_p
        vp, i32, i64, f32, u32 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint32
        L.pt_device_count.argtypes = [C.POINTER(C.c_int)]
        L.pt_device_name.argtypes = [i32, C.c_char_p, C.c_size_t]
        L.pt_scene_create.argtypes = [i32, C.POINTER(vp)]
        L.pt_scene_destroy.argtypes = [vp]
        L.pt_scene_add_material.argtypes = [vp, C.POINTER(MaterialDesc), C.POINTER(C.c_int)]
        L.pt_scene_add_mesh.argtypes = [vp, vp, vp, i32, vp, i32, vp, i32, i32, i32, C.POINTER(C.c_int)]
        L.pt_scene_add_sphere.argtypes = [vp, C.POINTER(C.c_float), f32, i32, C.POINTER(C.c_int)]
        L.pt_scene_set_hdri.argtypes = [vp, vp, i32, i32, f32, f32, f32, i32]
        L.pt_scene_commit.argtypes = [vp]
        L.pt_scene_clone.argtypes = [vp, i32, C.POINTER(vp)]
        L.pt_scene_save.argtypes = [vp, C.c_char_p]
        L.pt_scene_load.argtypes = [i32, C.c_char_p, C.POINTER(vp)]
        L.pt_octree_builds.argtypes = [C.POINTER(C.c_uint64)]
        L.pt_scene_octree_info.argtypes = [vp, i32, C.POINTER(OctreeInfo)]
        L.pt_scene_octree_dump.argtypes = [vp, i32, vp, i64, C.POINTER(C.c_int64)]
        L.pt_scene_device_bytes.argtypes = [vp] + [C.POINTER(C.c_uint64)] * 5
        L.pt_trace_rays.argtypes = [vp, vp, i64, vp, vp, i32, vp, vp]
        L.pt_camera_rays.argtypes = [i32, C.POINTER(Camera), u32, i32, vp, vp]
        L.pt_sample_hdri.argtypes = [vp, vp, i64, vp]
        L.pt_sample_material.argtypes = [vp, i32, vp, i64, vp, vp, vp, vp]
        L.pt_box_intersects_triangle.argtypes = [vp, vp, vp, vp, vp, i32, C.POINTER(C.c_int)]
        L.pt_child_tables.argtypes = [vp, vp]
        L.pt_render_accumulate.argtypes = [vp, C.POINTER(Camera), u32, i32, i32, i32, vp, vp, vp]
        L.pt_render_slices.argtypes = [C.POINTER(Camera), i32, i32, i32, C.POINTER(C.c_int)]
        L.pt_render_accumulate_tiles.argtypes = [vp, C.POINTER(Camera), u32, i32, i32, i32, vp, i32, i32, vp, vp]
        L.pt_render_accumulate_noalloc.argtypes = [vp, C.POINTER(Camera), u32, i32, i32, i32, vp, vp]
        L.pt_noalloc_max_partial_bytes.argtypes = []
        L.pt_noalloc_max_partial_bytes.restype = C.c_size_t
        L.pt_scene_set_launch.argtypes = [vp, i32, i32]
        L.pt_scene_get_launch.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.pt_scene_render_residency.argtypes = [vp, C.POINTER(C.c_int)]
        L.pt_scene_set_profiling.argtypes = [vp, i32]
        L.pt_scene_schedule_counters.argtypes = [vp, vp]
        L.pt_scene_wave_profile.argtypes = [vp, vp]
        L.pt_scene_last_render_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
        L.pt_compound_buffers.argtypes = [i32, vp, vp, i64, i32, vp]
        L.pt_reduce_sum.argtypes = [i32, vp, i32, C.POINTER(C.c_int), C.POINTER(vp), i64]
        L.pt_apply_filter.argtypes = [i32, vp, vp, i32, i32, vp, i32, i32, vp]
        L.pt_make_gaussian.argtypes = [C.c_uint, f32, vp]
        L.pt_make_mean.argtypes = [C.c_uint, vp]
        L.pt_normalize_image.argtypes = [i32, vp, i64, C.c_uint, vp]
        L.pt_apply_threshold.argtypes = [i32, vp, vp, i64, f32, vp]
        L.pt_image_add.argtypes = [i32, vp, vp, i64, f32, vp]
        L.pt_apply_aces.argtypes = [i32, vp, i64, vp]
        L.pt_gamma_correct.argtypes = [i32, vp, i64, f32, vp]
        L.pt_post_chain.argtypes = [i32, vp, vp, i32, i32, C.c_uint, vp]
        L.pt_malloc.argtypes = [i32, C.c_size_t, C.POINTER(vp)]
        L.pt_free.argtypes = [i32, vp]
        L.pt_memset.argtypes = [i32, vp, i32, C.c_size_t, vp]
        L.pt_upload.argtypes = [i32, vp, vp, C.c_size_t, vp]
        L.pt_download.argtypes = [i32, vp, vp, C.c_size_t, vp]
        L.pt_stream_synchronize.argtypes = [i32, vp]
        _lib = L
    return _lib


def source_hash():
    """SHA-256 over the library's sources (csrc/ and the public header): identifies the kernel a profile or a
    counter file was measured with.  Returns None when the sources are not next to the library."""
    import hashlib

    files = []
    csrc = os.path.join(PKG_ROOT, "csrc")
    hdr = os.path.join(os.path.dirname(PKG_ROOT), "include", "ptamd.h")
    if not os.path.isdir(csrc) or not os.path.exists(hdr):
        return None
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".h", ".cpp")) or name == "Makefile":
            files.append(os.path.join(csrc, name))
    files.append(hdr)
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def check(rc):
    if rc != 0:
        raise PtError(lib().pt_last_error().decode("utf-8", "replace"))


def device_count():
    n = C.c_int(0)
    check(lib().pt_device_count(C.byref(n)))
    return n.value


def make_camera(position, direction, res, focal_length=1.0, aperture=0.0, focus_distance=100.0, up=(0, 1, 0)):
    cam = Camera()
    cam.position[:] = [float(x) for x in position]
    cam.direction[:] = [float(x) for x in direction]
    cam.up[:] = [float(x) for x in up]
    cam.res_x, cam.res_y = int(res[0]), int(res[1])
    cam.focal_length, cam.aperture, cam.focus_distance = float(focal_length), float(aperture), float(focus_distance)
    return cam


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class DeviceBuffer:
    """Device memory obtained through the C ABI (pt_malloc); for callers without torch."""

    def __init__(self, device, nbytes):
        self.device, self.nbytes = device, int(nbytes)
        p = C.c_void_p()
        check(lib().pt_malloc(device, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def zero(self, stream=None):
        check(lib().pt_memset(self.device, self.ptr, 0, self.nbytes, stream))

    def upload(self, arr, stream=None):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        check(lib().pt_upload(self.device, self.ptr, arr.ctypes.data, arr.nbytes, stream))

    def download(self, shape, dtype=np.float32, stream=None):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        check(lib().pt_download(self.device, out.ctypes.data, self.ptr, out.nbytes, stream))
        return out

    def free(self):
        if self.ptr:
            check(lib().pt_free(self.device, self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Scene:
    """Mirror of the reference's Scene/Entity/Material host objects over the C ABI."""

    def __init__(self, device=0, _handle=None):
        self.L = lib()
        self.device = device
        if _handle is None:
            _handle = C.c_void_p()
            check(self.L.pt_scene_create(device, C.byref(_handle)))
        self.h = _handle
        self._keep = []

    def clone(self, device):
        """An uncommitted scene for another GPU that shares this one's host-side description (built octrees included)."""
        h = C.c_void_p()
        check(self.L.pt_scene_clone(self.h, device, C.byref(h)))
        return Scene(device, h)

    def save(self, path):
        """Writes the host-side description with its built octrees to a file (see load_scene)."""
        check(self.L.pt_scene_save(self.h, os.fsencode(path)))

    def close(self):
        if getattr(self, "h", None):
            self.L.pt_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_material(self, albedo=(0.8, 0.8, 0.8), specular=(0, 0, 0), emission=0.0, smoothness=0.0, smooth_shading=False,
                     maps=None):
        d = MaterialDesc()
        d.albedo[:] = [float(x) for x in albedo]
        d.specular[:] = [float(x) for x in specular]
        d.emission, d.smoothness, d.smooth_shading = float(emission), float(smoothness), int(bool(smooth_shading))
        for name, arr in (maps or {}).items():
            arr = np.ascontiguousarray(arr, dtype=np.uint8)
            self._keep.append(arr)
            setattr(d, name + "_map", arr.ctypes.data)
            setattr(d, name + "_w", arr.shape[1])
            setattr(d, name + "_h", arr.shape[0])
        out = C.c_int(-1)
        check(self.L.pt_scene_add_material(self.h, C.byref(d), C.byref(out)))
        return out.value

    def add_mesh(self, positions, tri, material, normals=None, uvs=None, octree_depth=0):
        positions = _f32(positions).reshape(-1, 3)
        tri = np.ascontiguousarray(tri, dtype=np.int32).reshape(-1, 6)
        nrm = _f32(normals).reshape(-1, 3) if normals is not None else None
        uv = _f32(uvs).reshape(-1, 2) if uvs is not None and len(uvs) else None
        out = C.c_int(-1)
        check(self.L.pt_scene_add_mesh(self.h, positions.ctypes.data, nrm.ctypes.data if nrm is not None else None,
                                       len(positions), tri.ctypes.data, len(tri), uv.ctypes.data if uv is not None else None,
                                       len(uv) if uv is not None else 0, material, int(octree_depth), C.byref(out)))
        return out.value

    def add_sphere(self, center, radius, material):
        c = (C.c_float * 3)(*[float(x) for x in center])
        out = C.c_int(-1)
        check(self.L.pt_scene_add_sphere(self.h, c, float(radius), material, C.byref(out)))
        return out.value

    def set_hdri(self, rgb, exposure=1.0, contrast=1.0, rot=0.0, smooth=False):
        rgb = _f32(rgb)
        check(self.L.pt_scene_set_hdri(self.h, rgb.ctypes.data, rgb.shape[1], rgb.shape[0], float(exposure), float(contrast),
                                       float(rot), int(bool(smooth))))

    def commit(self):
        check(self.L.pt_scene_commit(self.h))

    def octree_info(self, entity):
        info = OctreeInfo()
        check(self.L.pt_scene_octree_info(self.h, entity, C.byref(info)))
        per = np.array(list(info.nodes_per_depth)[: info.max_depth + 1], dtype=np.int64)
        return dict(max_depth=info.max_depth, n_nodes=info.n_nodes, n_leaf_refs=info.n_leaf_refs, max_leaf=info.max_leaf,
                    nodes_per_depth=per)

    def octree_dump(self, entity):
        n = C.c_int64(0)
        check(self.L.pt_scene_octree_dump(self.h, entity, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=np.int32)
        check(self.L.pt_scene_octree_dump(self.h, entity, out.ctypes.data, n.value, C.byref(n)))
        return out

    def device_bytes(self):
        v = [C.c_uint64(0) for _ in range(5)]
        check(self.L.pt_scene_device_bytes(self.h, *[C.byref(x) for x in v]))
        return dict(zip(["nodes", "tri_records", "shade_records", "textures", "hdri"], [int(x.value) for x in v]))

    def trace(self, rays, log_stride=0, want_counters=True):
        """rays: host (n,6) float32 -> hits (structured array), log or None, counters dict."""
        rays = _f32(rays).reshape(-1, 6)
        hits = np.zeros(len(rays), dtype=HIT_DTYPE)
        log = np.zeros((len(rays), log_stride), dtype=np.int32) if log_stride else None
        cnt = Counters()
        check(self.L.pt_trace_rays(self.h, rays.ctypes.data, len(rays), hits.ctypes.data,
                                   log.ctypes.data if log is not None else None, log_stride,
                                   C.byref(cnt) if want_counters else None, None))
        return hits, log, cnt.as_dict()

    def trace_device(self, d_rays_ptr, n, d_hits_ptr, stream=None):
        check(self.L.pt_trace_rays(self.h, d_rays_ptr, n, d_hits_ptr, None, 0, None, stream))

    def sample_hdri(self, dirs):
        dirs = _f32(dirs).reshape(-1, 3)
        out = np.zeros_like(dirs)
        check(self.L.pt_sample_hdri(self.h, dirs.ctypes.data, len(dirs), out.ctypes.data))
        return out

    def sample_material(self, material, uv):
        """-> albedo (n,3), specular (n,3), smoothness (n,), tangent-space normal (n,3)"""
        uv = _f32(uv).reshape(-1, 2)
        n = len(uv)
        a, s, nn = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
        sm = np.zeros(n, np.float32)
        check(self.L.pt_sample_material(self.h, material, uv.ctypes.data, n, a.ctypes.data, s.ctypes.data, sm.ctypes.data, nn.ctypes.data))
        return a, s, sm, nn

    def render_accumulate(self, cam, seed, sample_begin, sample_end, n_split, d_split_ptr, want_counters=False, stream=None,
                          tiles=(0, 1)):
        """tiles = (rank, ranks): only the rank-th of `ranks` interleaved shares of the 8x8-pixel tiles (pt_render_accumulate_tiles)."""
        cnt = Counters()
        check(self.L.pt_render_accumulate_tiles(self.h, C.byref(cam), seed, sample_begin, sample_end, n_split, d_split_ptr,
                                                int(tiles[0]), int(tiles[1]), C.byref(cnt) if want_counters else None, stream))
        return cnt.as_dict() if want_counters else None

    def render_accumulate_noalloc(self, cam, seed, sample_begin, sample_end, n_split, d_split_ptr, stream=None):
        """pt_render_accumulate without any allocation (fixed scratch array; see include/ptamd.h for its size limit)."""
        check(self.L.pt_render_accumulate_noalloc(self.h, C.byref(cam), seed, sample_begin, sample_end, n_split, d_split_ptr, stream))

    def set_launch(self, kernel_build=None, quorum=None):
        """Fixes the kernel build ("heavy" / "light") and / or the quorum (1..64); None leaves a choice as it is."""
        kb = -1 if kernel_build is None else {"light": 0, "heavy": 1}[kernel_build]
        check(self.L.pt_scene_set_launch(self.h, kb, 0 if quorum is None else int(quorum)))

    def launch(self):
        kb, q = C.c_int(0), C.c_int(0)
        check(self.L.pt_scene_get_launch(self.h, C.byref(kb), C.byref(q)))
        return dict(kernel_build="heavy" if kb.value else "light", quorum=int(q.value))

    def waves_per_simd(self):
        """Workgroups of the render kernel resident per CU (= waves per SIMD the kernel was instantiated for)."""
        n = C.c_int(0)
        check(self.L.pt_scene_render_residency(self.h, C.byref(n)))
        return int(n.value)

    def schedule_counters(self):
        out = (C.c_uint64 * 8)()
        check(self.L.pt_scene_schedule_counters(self.h, out))
        names = ["ent", "node", "leaf", "front"]
        v = [int(x) for x in out]
        return {n: dict(lanes=v[i], trips=v[4 + i], occupancy=(v[i] / (64.0 * v[4 + i]) if v[4 + i] else 0.0)) for i, n in enumerate(names)}

    def block_cycles(self):
        """Shares of the waves' life time in the last counted launch: entity / node / leaf blocks, the geometry scheduler
        around them, and the shade / refill round (pt_scene_wave_profile)."""
        out = (C.c_uint64 * 11)()
        check(self.L.pt_scene_wave_profile(self.h, out))
        ent, node, leaf, trace, total = [int(x) for x in out[:5]]
        self._idle_lanes = dict(finished=int(out[5]), at_leaf=int(out[6]), other=int(out[7]))
        self._span = dict(start_after_first=int(out[8]), end_before_last=int(out[9]), span=int(out[10]))
        if total == 0:
            return {}
        return dict(ent=ent / total, node=node / total, leaf=leaf / total, scheduler=(trace - ent - node - leaf) / total,
                    round=(total - trace) / total, wave_cycles=total)

    def launch_span(self):
        """In ticks of the chip-wide 100 MHz clock, for the last counted launch: mean delay of a wave's start after the first wave's, mean lead of a wave's
        end before the last wave's (the tail), and the span from the first start to the last end (call block_cycles first)."""
        return dict(self._span)

    def node_trip_idle_lanes(self):
        """Lanes that sat out the node trips of the last counted launch, by reason (sums over all trips; call block_cycles first)."""
        return dict(self._idle_lanes)

    def set_profiling(self, enable=True):
        check(self.L.pt_scene_set_profiling(self.h, int(bool(enable))))

    def last_render_kernel_ms(self):
        ms = C.c_float(0)
        check(self.L.pt_scene_last_render_kernel_ms(self.h, C.byref(ms)))
        return float(ms.value)


def load_scene(path, device=0, commit=True):
    """A Scene from a file written by Scene.save: nothing is rebuilt."""
    h = C.c_void_p()
    check(lib().pt_scene_load(device, os.fsencode(path), C.byref(h)))
    sc = Scene(device, h)
    if commit:
        sc.commit()
    return sc


def octree_builds():
    """Octrees this process has built since the library was loaded."""
    n = C.c_uint64(0)
    check(lib().pt_octree_builds(C.byref(n)))
    return int(n.value)


def scene_from_description(desc, device=0, commit=True):
    """Builds a Scene (and its Camera) from a plain-array scene description (see scenes.py)."""
    sc = Scene(device)
    for m in desc["materials"]:
        sc.add_material(m["albedo"], m["specular"], m["emission"], m["smoothness"], m["smooth_shading"], m["maps"])
    for e in desc["entities"]:
        if e["kind"] == "mesh":
            sc.add_mesh(e["positions"], e["tri"], e["material"], normals=e["normals"], uvs=e["uvs"], octree_depth=e["octree_depth"])
        else:
            sc.add_sphere(e["center"], e["radius"], e["material"])
    if desc.get("hdri"):
        hd = desc["hdri"]
        sc.set_hdri(hd["rgb"], hd["exposure"], hd["contrast"], hd["rot"], hd["smooth"])
    if commit:
        sc.commit()
    c = desc["camera"]
    cam = make_camera(c["position"], c["direction"], c["res"], c["focal_length"], c["aperture"], c["focus_distance"])
    return sc, cam


def render_slices(cam, sample_begin, sample_end, n_split=4):
    """Sample slices (= partial images of the workspace) of a pt_render_accumulate call with these arguments."""
    n = C.c_int(0)
    check(lib().pt_render_slices(C.byref(cam), sample_begin, sample_end, n_split, C.byref(n)))
    return int(n.value)


def camera_rays(cam, seed, sample, device=0):
    out = np.zeros((cam.res_x * cam.res_y, 6), dtype=np.float32)
    check(lib().pt_camera_rays(device, C.byref(cam), seed, sample, out.ctypes.data, None))
    return out


def box_intersects_triangle(bmin, bmax, v0, v1, v2, pad_box=False):
    a = [_f32(x) for x in (bmin, bmax, v0, v1, v2)]
    out = C.c_int(0)
    check(lib().pt_box_intersects_triangle(*[x.ctypes.data for x in a], int(pad_box), C.byref(out)))
    return bool(out.value)


def make_gaussian(size, sigma):
    t = np.zeros((2 * size + 1, 2 * size + 1), np.float32)
    check(lib().pt_make_gaussian(size, float(sigma), t.ctypes.data))
    return t


def make_mean(size):
    t = np.zeros((2 * size + 1, 2 * size + 1), np.float32)
    check(lib().pt_make_mean(size, t.ctypes.data))
    return t


def render_image(scene, cam, spp, seed=1337, n_split=4, sample_begin=0, device=0, want_counters=False, tiles=(0, 1)):
    """Convenience: renders sample_begin..sample_begin+spp-1 (of the tile share `tiles`, default every tile) and returns the
    un-normalised HxWx3 sum (host)."""
    n_pixels = cam.res_x * cam.res_y
    split = DeviceBuffer(device, n_split * n_pixels * 12)
    out = DeviceBuffer(device, n_pixels * 12)
    split.zero()
    cnt = scene.render_accumulate(cam, seed, sample_begin, sample_begin + spp, n_split, split.ptr, want_counters, tiles=tiles)
    check(lib().pt_compound_buffers(device, split.ptr, out.ptr, n_pixels, n_split, None))
    img = out.download((cam.res_y, cam.res_x, 3))
    split.free()
    out.free()
    return (img, cnt) if want_counters else img


def apply_filter_host(img, taps, bilateral=False, device=0):
    img = _f32(img)
    h, w = img.shape[0], img.shape[1]
    taps = _f32(taps)
    ks = (taps.shape[0] - 1) // 2
    d_in, d_out = DeviceBuffer(device, img.nbytes), DeviceBuffer(device, img.nbytes)
    d_in.upload(img)
    check(lib().pt_apply_filter(device, d_in.ptr, d_out.ptr, w, h, taps.ctypes.data, ks, 1 if bilateral else 0, None))
    out = d_out.download(img.shape)
    d_in.free()
    d_out.free()
    return out


def post_chain_host(img_sum, n_samples, device=0):
    img_sum = _f32(img_sum)
    h, w = img_sum.shape[0], img_sum.shape[1]
    d_in, d_out = DeviceBuffer(device, img_sum.nbytes), DeviceBuffer(device, img_sum.nbytes)
    d_in.upload(img_sum)
    check(lib().pt_post_chain(device, d_in.ptr, d_out.ptr, w, h, int(n_samples), None))
    out = d_out.download(img_sum.shape)
    d_in.free()
    d_out.free()
    return out
