"""Writes the procedural stand-in assets of the north-star scene as files, under the paths the
reference's scene script names (src/main.cu:17-19,24,45): the statue mesh as OBJ, the three texture
maps as binary PPM/PGM (the reference reads JPEG through stb_image, which is not re-created), and
the environment map as a Radiance .hdr.  `python -m ptamd.export_assets <examples dir> [--small]`.

Floats are written with 9 significant digits, so a reader that parses decimal text correctly gets the
same fp32 values back."""
import argparse
import os

import numpy as np

from . import scenes


def write_obj(path, positions, tri, uvs=None):
    with open(path, "w") as f:
        f.write("# procedural stand-in mesh\no mesh\n")
        for p in positions:
            f.write("v %.9g %.9g %.9g\n" % (p[0], p[1], p[2]))
        if uvs is not None:
            for t in uvs:
                f.write("vt %.9g %.9g\n" % (t[0], t[1]))
        for t in tri:
            if uvs is not None:
                f.write("f %d/%d %d/%d %d/%d\n" % (t[0] + 1, t[3] + 1, t[1] + 1, t[4] + 1, t[2] + 1, t[5] + 1))
            else:
                f.write("f %d %d %d\n" % (t[0] + 1, t[1] + 1, t[2] + 1))


def write_pnm(path, texels):
    """uint8 (h, w, 3) -> P6, (h, w) -> P5"""
    texels = np.ascontiguousarray(texels, dtype=np.uint8)
    h, w = texels.shape[:2]
    magic = b"P6" if texels.ndim == 3 else b"P5"
    with open(path, "wb") as f:
        f.write(magic + b"\n%d %d\n255\n" % (w, h))
        f.write(texels.tobytes())


def rgbe_encode(rgb):
    """float (h, w, 3) -> uint8 (h, w, 4), the shared-exponent encoding of the Radiance format"""
    rgb = np.asarray(rgb, dtype=np.float32)
    m = rgb.max(axis=2)
    mant, exp = np.frexp(m.astype(np.float64))
    scale = np.where(m > 1e-32, mant * 256.0 / np.maximum(m, 1e-38), 0.0)
    out = np.zeros(rgb.shape[:2] + (4,), dtype=np.uint8)
    out[..., :3] = np.clip(rgb * scale[..., None], 0, 255).astype(np.uint8)
    out[..., 3] = np.where(m > 1e-32, exp + 128, 0).astype(np.uint8)
    return out


def rgbe_decode(rgbe):
    """the decoding of external/hdrloader.h:54-59: mantissa / 256 * 2^(e - 128), in fp32"""
    e = rgbe[..., 3].astype(np.int32) - 128
    d = np.ldexp(np.float64(1.0), e).astype(np.float32)
    return ((rgbe[..., :3].astype(np.float32) / np.float32(256.0)) * d[..., None]).astype(np.float32)


def write_hdr(path, rgb, rle=False):
    rgbe = rgbe_encode(rgb)
    h, w = rgbe.shape[:2]
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w))
        if not rle:
            f.write(rgbe.tobytes())
            return rgbe
        for y in range(h):      # new-style scanlines, every component as literal runs of <= 128
            f.write(bytes([2, 2, (w >> 8) & 255, w & 255]))
            for c in range(4):
                row = rgbe[y, :, c].tobytes()
                for x in range(0, w, 128):
                    chunk = row[x:x + 128]
                    f.write(bytes([len(chunk)]) + chunk)
    return rgbe


def export(root, small=False):
    n_tris, tex_res, hdri_res = (8000, 256, (512, 256)) if small else (80000, 4096, (4096, 2048))
    os.makedirs(os.path.join(root, "hcandersen", "source"), exist_ok=True)
    os.makedirs(os.path.join(root, "hcandersen", "textures"), exist_ok=True)
    os.makedirs(os.path.join(root, "hdris"), exist_ok=True)
    pos, faces, uv = scenes.statue_mesh(n_tris)
    write_obj(os.path.join(root, "hcandersen", "source", "80k.obj"), pos, scenes.tri6(faces, faces), uv)
    albedo, normal, gloss = scenes.procedural_textures(tex_res)
    write_pnm(os.path.join(root, "hcandersen", "textures", "diff.ppm"), albedo)
    write_pnm(os.path.join(root, "hcandersen", "textures", "normal.ppm"), normal)
    write_pnm(os.path.join(root, "hcandersen", "textures", "gloss.pgm"), gloss)
    write_hdr(os.path.join(root, "hdris", "quattro_canti_4k.hdr"), scenes.procedural_hdri(hdri_res[0], hdri_res[1]))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("root")
    ap.add_argument("--small", action="store_true")
    a = ap.parse_args()
    export(a.root, a.small)
