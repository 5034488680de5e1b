"""Diagnostic: which (pixel, sample) paths differ between the HIP render and the oracle, and by how much.
usage: python profiles/diag/path_diff.py [workload] [W] [H] [spp]"""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "cuda-path-tracer_amd", "python")); sys.path.insert(0, os.path.join(REPO, "oracle"))
import oracle_api as O, ptamd
from ptamd import scenes as S
name = sys.argv[1] if len(sys.argv) > 1 else "teapot"
W, H, spp = (int(x) for x in (sys.argv[2:5] if len(sys.argv) > 4 else (96, 54, 64)))
kw = dict(hcandersen=dict(n_tris=8000, tex_res=256, hdri_res=(512, 256))).get(name, {})
desc = S.CONFIGS[name]["builder"](res=(W, H), **kw)
g, cam = ptamd.scene_from_description(desc)
c, ocam = O.scene_from_description(desc)
tot_g = np.zeros((H, W, 3)); tot_c = np.zeros((H, W, 3))
n_diff = 0; big = []
for s in range(spp):
    a = ptamd.render_image(g, cam, 1, seed=1337, n_split=1, sample_begin=s).astype(np.float64)
    b, _ = c.render(ocam, 1337, s, s + 1, n_threads=8); b = b.astype(np.float64)
    tot_g += a; tot_c += b
    d = np.abs(a - b).max(axis=-1)
    rel = d / np.maximum(np.abs(b).max(axis=-1), 1e-3)
    idx = np.argwhere(rel > 1e-3)
    n_diff += len(idx)
    for (y, x) in idx[:50]:
        big.append((float(d[y, x]), s, int(y), int(x), a[y, x].tolist(), b[y, x].tolist()))
big.sort(reverse=True)
rms = np.sqrt(np.mean((tot_c / spp) ** 2)); err = np.sqrt(np.mean(((tot_g - tot_c) / spp) ** 2))
print(f"{name} {W}x{H} {spp}spp: paths differing by > 1e-3 rel: {n_diff} of {W*H*spp} ({n_diff/(W*H*spp):.2e}); rmse/rms = {err/rms:.4f}")
for row in big[:25]:
    print("  diff %.4f sample %d pixel (%d,%d) gpu %s cpu %s" % (row[0], row[1], row[2], row[3], np.round(row[4], 4).tolist(), np.round(row[5], 4).tolist()))
h = np.histogram([r[0] for r in big], bins=[0, 1e-3, 1e-2, 1e-1, 1, 10, 100])
print("  |diff| histogram of listed paths", h[0].tolist(), h[1].tolist())
