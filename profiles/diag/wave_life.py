"""How much of a launch do the waves of the render kernel live through?  The counting build stamps every wave's first and last
instruction with the shader clock; sum / (waves x kernel time) against the same ratio of a long launch shows the share of a short
launch in which waves have retired (the tail) or not started.  python profiles/diag/wave_life.py [workload]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "cuda-path-tracer_amd", "python"))
import numpy as np
import torch
torch.cuda.init()
import ptamd
from ptamd import scenes as S

w = sys.argv[1] if len(sys.argv) > 1 else "hcandersen"
cfg = S.CONFIGS[w]
res = cfg["res"]
scene, cam = ptamd.scene_from_description(cfg["builder"](res=res), device=0)
split = torch.zeros((4, res[1], res[0], 3), dtype=torch.float32, device="cuda:0")
stream = torch.cuda.current_stream().cuda_stream
scene.set_profiling(True)
n_waves = 256 * 4 * 4
for spp in ((32, 64, 128, 256, 1024) if len(sys.argv) < 3 else [int(x) for x in sys.argv[2:]]):
    for counted in (False, True):
        split.zero_()
        scene.render_accumulate(cam, 1337, 0, spp, 4, split.data_ptr(), counted, stream)
        torch.cuda.synchronize()
        ms = scene.last_render_kernel_ms()
        if counted:
            bc = scene.block_cycles()
            sp = scene.launch_span()
            clk = 100e3                    # ticks per ms
            print(f"{w} spp {spp}: counting kernel {ms:.2f} ms; mean start {sp['start_after_first'] / clk:.3f} ms after the first wave, "
                  f"mean end {sp['end_before_last'] / clk:.3f} ms before the last, span {sp['span'] / clk:.3f} ms", flush=True)
        else:
            print(f"{w} spp {spp}: timed kernel {ms:.2f} ms", flush=True)
