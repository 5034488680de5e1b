import os, sys
sys.path.insert(0, "cuda-path-tracer_amd/python")
import ptamd
from ptamd import scenes as S
desc = S.hcandersen_scene()
sc, cam = ptamd.scene_from_description(desc)
n = cam.res_x * cam.res_y
split = ptamd.DeviceBuffer(0, 4 * n * 12); split.zero()
for spp in (32, 256):
    sc.render_accumulate(cam, 1337, 0, spp, 4, split.ptr, True)
    c = sc.schedule_counters()
    tmin = (1 << 62) - c["ent"]["trips"]; tmax = c["node"]["trips"]; life = c["leaf"]["trips"]; nw = c["front"]["trips"]
    print(f"spp {spp}: kernel span {(tmax - tmin) / 100e3:.3f} ms (counting build), waves {nw}, mean wave lifetime {life / nw / 100e3:.3f} ms")
