"""Scene files: a JSON front end that replaces the reference's edit-main.cu-and-recompile workflow
(src/main.cu:7-69).  Every key maps to one call of the reference's scene script:

  {"materials": {"hc": {"albedo": [0.8, 0.8, 0.8], "specular": [0.2, 0.2, 0.2], "emission": 0, "smoothness": 1.0,
                        "smooth_shading": false, "albedo_map": "diff.ppm", "normal_map": "normal.ppm",
                        "specular_map": null, "smoothness_map": "gloss.pgm"}},
   "entities":  [{"obj": "80k.obj", "material": "hc", "octree": true, "normals": "reference",
                  "transforms": [{"rotate": [-1.5707963, 1.5707963, 0]}, {"translate": [0, 22.65, 0]}]},
                 {"sphere": {"center": [10, 40, 30], "radius": 3.0}, "material": "light"}],
   "hdri":      {"file": "sky.hdr", "exposure": 1.0, "contrast": 1.0, "rotate": 0.85, "smooth": true},
   "camera":    {"position": [40.5, 30.5, -50], "direction": [-0.5, 0, 1], "resolution": [1920, 1080],
                 "focal_length": 2.5, "aperture": 1.0, "focus_distance": 85},
   "render":    {"spp": 2048, "seed": 1337}}

Paths are relative to the scene file.  "octree": true builds the reference's depth-6 tree, an integer
picks another depth, false / absent leaves the mesh brute force.  Transforms are applied in order
with the reference's semantics (scale and rotate about the entity centre, translate moves it).
"normals": "reference" keeps the reference's handling of OBJ normals (vn indexed by VERTEX id when
the file has any -- the known flaw of src/entity.cu:58-65 -- else the face-normal average),
"computed" always uses the face-normal average, "file" averages the vn each vertex is given by its
face corners.
"""
import hashlib
import json
import os

import numpy as np

from . import scenes as S


def read_pnm(path):
    """binary P6 / P5 with maxval 255 -> uint8 (h, w, 3) / (h, w)"""
    with open(path, "rb") as f:
        data = f.read()
    tokens, pos = [], 0
    while len(tokens) < 4:
        while data[pos:pos + 1].isspace():
            pos += 1
        if data[pos:pos + 1] == b"#":
            pos = data.index(b"\n", pos) + 1
            continue
        end = pos
        while not data[end:end + 1].isspace():
            end += 1
        tokens.append(data[pos:end])
        pos = end
    pos += 1
    magic, w, h, maxval = tokens[0], int(tokens[1]), int(tokens[2]), int(tokens[3])
    if magic not in (b"P6", b"P5") or maxval != 255:
        raise ValueError(f"{path}: only binary PPM/PGM with maxval 255 are read")
    ch = 3 if magic == b"P6" else 1
    a = np.frombuffer(data, np.uint8, w * h * ch, pos)
    return a.reshape(h, w, 3) if ch == 3 else a.reshape(h, w)


def read_hdr(path):
    """Radiance RGBE -> float32 (h, w, 3), first scanline first; flat and new-style RLE scanlines;
    component = mantissa / 256 * 2^(e - 128) (external/hdrloader.h:54-59)."""
    from .export_assets import rgbe_decode

    with open(path, "rb") as f:
        data = f.read()
    if not data.startswith(b"#?"):
        raise ValueError(f"{path}: not a Radiance file")
    pos = data.index(b"\n\n") + 2
    end = data.index(b"\n", pos)
    dims = data[pos:end].split()
    if len(dims) != 4 or dims[0] != b"-Y" or dims[2] != b"+X":
        raise ValueError(f"{path}: unsupported orientation {data[pos:end]!r}")
    h, w = int(dims[1]), int(dims[3])
    pos = end + 1
    buf = np.frombuffer(data, np.uint8)
    rgbe = np.zeros((h, w, 4), np.uint8)
    if len(data) - pos == w * h * 4 and not (8 <= w < 32768 and buf[pos] == 2 and buf[pos + 1] == 2):
        rgbe[:] = buf[pos:pos + w * h * 4].reshape(h, w, 4)
        return rgbe_decode(rgbe)
    for y in range(h):
        if 8 <= w < 32768 and buf[pos] == 2 and buf[pos + 1] == 2 and not (buf[pos + 2] & 128):
            pos += 4
            for c in range(4):
                x = 0
                while x < w:
                    code = int(buf[pos]); pos += 1
                    if code > 128:
                        n = code & 127
                        rgbe[y, x:x + n, c] = buf[pos]; pos += 1
                    else:
                        n = code
                        rgbe[y, x:x + n, c] = buf[pos:pos + n]; pos += n
                    x += n
        else:
            rgbe[y] = buf[pos:pos + 4 * w].reshape(w, 4)
            pos += 4 * w
    return rgbe_decode(rgbe)


def face_normal_average(pos, tri):
    """src/entity.cu:110-133: area-unweighted average of the normalised face normals, fp32, in face order."""
    f = np.float32
    n = np.zeros_like(pos)
    a, b, c = tri[:, 0], tri[:, 1], tri[:, 2]
    e1 = (pos[b] - pos[a]).astype(f)
    e2 = (pos[c] - pos[a]).astype(f)
    cr = np.stack([e1[:, 1] * e2[:, 2] - e1[:, 2] * e2[:, 1], e1[:, 2] * e2[:, 0] - e1[:, 0] * e2[:, 2],
                   e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]], axis=1).astype(f)
    t = S._normalize_rows(cr)
    for k in range(len(tri)):       # sequential: the accumulation order is part of the result
        n[a[k]] += t[k]
        n[b[k]] += t[k]
        n[c[k]] += t[k]
    with np.errstate(invalid="ignore", divide="ignore"):
        return S._normalize_rows(n)


def _load_obj_with_normal_indices(path):
    """load_obj plus the vn index of every face corner (-1 when absent)"""
    pos, tri, uvs, vns = S.load_obj(path)
    ni = []
    with open(path, "r") as f:
        for line in f:
            line = line.split("#", 1)[0].strip()
            if line.startswith("f "):
                row = []
                for p in line.split()[1:4]:
                    fields = p.split("/")
                    if len(fields) > 2 and fields[2] != "":
                        v = int(fields[2])
                        row.append(v - 1 if v > 0 else len(vns) + v)
                    else:
                        row.append(-1)
                ni.append(row)
    return pos, tri, uvs, vns, np.asarray(ni, dtype=np.int32).reshape(-1, 3)


def _mesh_normals(mode, pos, tri, vns, ni):
    if mode == "reference":
        if len(vns) == 0:
            return None                                   # the library computes the face-normal average
        n = np.zeros_like(pos)
        m = min(len(pos), len(vns))
        n[:m] = S._normalize_rows(vns[:m])                # vn indexed by vertex id, like the reference
        return n
    if mode == "computed":
        return None
    if mode == "file":
        if len(vns) == 0 or (ni < 0).any():
            raise ValueError("normals: 'file' needs a vn index on every face corner")
        n = np.zeros_like(pos)
        for k in range(3):
            np.add.at(n, tri[:, k], vns[ni[:, k]])
        with np.errstate(invalid="ignore", divide="ignore"):
            return S._normalize_rows(n)
    raise ValueError(f"unknown normals mode {mode!r}")


def load_scene(path):
    """Scene file -> (scene description for ptamd.scene_from_description, render settings)"""
    base = os.path.dirname(os.path.abspath(path))
    with open(path) as f:
        doc = json.load(f)

    def rel(p):
        return p if os.path.isabs(p) else os.path.join(base, p)

    names, mats = {}, []
    for name, m in doc.get("materials", {}).items():
        maps = {}
        for key in ("albedo", "specular", "normal", "smoothness"):
            p = m.get(key + "_map")
            if p:
                tex = read_pnm(rel(p))
                if key == "smoothness":
                    tex = tex if tex.ndim == 2 else np.ascontiguousarray(tex[..., 0])
                elif tex.ndim == 2:
                    tex = np.repeat(tex[..., None], 3, axis=2)
                maps[key] = np.ascontiguousarray(tex)
        names[name] = len(mats)
        mats.append(S._mat(m.get("albedo", (0.8, 0.8, 0.8)), m.get("specular", (0, 0, 0)), m.get("emission", 0.0),
                           m.get("smoothness", 0.0), m.get("smooth_shading", False), maps))
    ents = []
    for e in doc.get("entities", []):
        if e["material"] not in names:
            raise ValueError(f"unknown material {e['material']!r}")
        mat = names[e["material"]]
        if "sphere" in e:
            center = np.asarray(e["sphere"]["center"], np.float32)
            radius = np.float32(e["sphere"]["radius"])
            for t in e.get("transforms", []):
                if "scale" in t:
                    radius = np.float32(radius * np.float32(t["scale"]))
                elif "translate" in t:
                    center = (center + np.asarray(t["translate"], np.float32)).astype(np.float32)
            ents.append(S._sphere(center, radius, mat))
            continue
        pos, tri, uvs, vns, ni = _load_obj_with_normal_indices(rel(e["obj"]))
        mode = e.get("normals", "reference")
        normals = _mesh_normals(mode, pos, tri, vns, ni)
        center = S.mesh_center(pos)
        for t in e.get("transforms", []):
            if "scale" in t:
                pos = S.scale_mesh(pos, center, t["scale"])
            elif "rotate" in t:
                if normals is None:        # rotation also turns the normals: they have to exist first
                    normals = face_normal_average(pos, tri)
                pos, normals = S.rotate_mesh(pos, normals, center, t["rotate"])
            elif "translate" in t:
                pos, center = S.translate_mesh(pos, center, t["translate"])
            else:
                raise ValueError(f"unknown transform {t!r}")
        oc = e.get("octree", False)
        depth = 6 if oc is True else int(oc or 0)
        ents.append(S._mesh(pos, tri, mat, normals=normals, uvs=uvs if len(uvs) else None, octree_depth=depth))
    hdri = None
    if doc.get("hdri"):
        h = doc["hdri"]
        hdri = dict(rgb=read_hdr(rel(h["file"])), exposure=float(h.get("exposure", 1.0)), contrast=float(h.get("contrast", 1.0)),
                    rot=float(h.get("rotate", 0.0)) % 1.0, smooth=bool(h.get("smooth", False)))
    c = doc["camera"]
    cam = S._camera(c["position"], c["direction"], c["resolution"], c.get("focal_length", 1.0), c.get("aperture", 0.0),
                    c.get("focus_distance", 100.0))
    r = doc.get("render", {})
    settings = dict(spp=int(r.get("spp", 1024)), seed=int(r.get("seed", 1337)))
    return dict(name=os.path.basename(path), materials=mats, entities=ents, hdri=hdri, camera=cam), settings


def description_hash(desc):
    """SHA-256 over everything of a scene description that a render depends on -- the arrays as they were loaded (mesh
    vertices, normals, triangles, uvs, texture maps, environment map), the materials' and entities' scalar fields and the
    camera -- so that it changes when a mesh, texture or environment file the scene file names is edited, not only when
    the scene file itself is."""
    h = hashlib.sha256()

    def put(tag, v):
        h.update(tag.encode())
        if isinstance(v, dict):
            for k in sorted(v):
                put(tag + "." + str(k), v[k])
        elif isinstance(v, (list, tuple)) and any(isinstance(x, (dict, np.ndarray, list, tuple)) for x in v):
            for i, x in enumerate(v):
                put(f"{tag}[{i}]", x)
        elif v is None:
            h.update(b"none")
        else:
            a = np.ascontiguousarray(v)
            h.update(str(a.dtype).encode() + str(a.shape).encode())
            h.update(a.tobytes())

    put("scene", desc)
    return h.hexdigest()
