"""Kernel time of the render kernel against the launch length (samples per pixel) on the north-star scene.
usage: python profiles/diag/launch_len.py [spp ...]"""
import os, sys, json
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "cuda-path-tracer_amd", "python"))
import ptamd
from ptamd import scenes as S
spps = [int(x) for x in sys.argv[1:]] or [1, 4, 8, 16, 32, 64, 128, 256]
desc = S.hcandersen_scene()
sc, cam = ptamd.scene_from_description(desc)
n = cam.res_x * cam.res_y
split = ptamd.DeviceBuffer(0, 4 * n * 12)
split.zero()
sc.set_profiling(True)
L = ptamd.lib()
for spp in spps:
    ms = []
    for rep in range(3):
        sc.render_accumulate(cam, 1337, 0, spp, 4, split.ptr)
        ms.append(sc.last_render_kernel_ms())
    print(f"spp {spp:5d}  kernel ms {min(ms):9.3f}  per spp {min(ms)/spp:7.4f}  Mpaths/s {n*spp/min(ms)/1e3:8.1f}", flush=True)
