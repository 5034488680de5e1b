"""Experiment: raw closest-hit throughput of pt_trace_rays on primary and diffuse-bounce rays of the
hcandersen-class scene (device-resident buffers), to compare with the render kernel's segment rate."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cuda-path-tracer_amd", "python"))
import numpy as np
import torch
import ptamd
from ptamd import scenes as S

desc = S.hcandersen_scene(tex_res=512, hdri_res=(512, 256))
scene, cam = ptamd.scene_from_description(desc, device=0)
n = cam.res_x * cam.res_y
rays = torch.empty((n, 6), dtype=torch.float32, device="cuda")
import ctypes as C
L = ptamd.lib()
ptamd.check(L.pt_camera_rays(0, C.byref(cam), 1337, 0, rays.data_ptr(), None))
hits = torch.empty((n, 13), dtype=torch.float32, device="cuda")

def run(r, label, reps=5):
    m = r.shape[0]
    h = torch.empty((m, 13), dtype=torch.float32, device="cuda")
    scene.trace_device(r.data_ptr(), m, h.data_ptr())
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        scene.trace_device(r.data_ptr(), m, h.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    ent = h[:, 0].view(torch.int32)
    print(f"{label}: {m/dt/1e6:.1f} Mrays/s  ({dt*1e3:.2f} ms, hit fraction {(ent >= 0).float().mean().item():.3f})", flush=True)
    return h

h = run(rays, "primary")
ent = h[:, 0].view(torch.int32)
sel = ent >= 0
pos = h[sel][:, 7:10]
nrm = h[sel][:, 10:13]
g = torch.Generator(device="cuda"); g.manual_seed(1)
d = torch.randn((pos.shape[0], 3), device="cuda", generator=g)
d = d / d.norm(dim=1, keepdim=True)
d = torch.where(((d * nrm).sum(1, keepdim=True) < 0), -d, d)
sec = torch.cat([pos + 0.001 * nrm, d], dim=1).contiguous()
reps = max(1, n // sec.shape[0])
sec = sec.repeat(reps, 1)[:n].contiguous()
# shuffle so that a wave holds unrelated rays, like bounce rays of different pixels
perm = torch.randperm(sec.shape[0], device="cuda", generator=g)
run(sec, "secondary (coherent order)")
run(sec[perm].contiguous(), "secondary (shuffled)")
mix = torch.cat([rays[: n // 3], sec[perm][: 2 * n // 3]], dim=0)
mix = mix[torch.randperm(mix.shape[0], device="cuda", generator=g)].contiguous()
run(mix, "mix 1/3 primary + 2/3 secondary, shuffled")
