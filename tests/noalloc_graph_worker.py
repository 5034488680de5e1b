"""Worker of tests/test_zz_no_alloc.py: pt_render_accumulate_noalloc allocates no device memory and can be captured in
a HIP graph.  Runs in a process of its own because torch (used for the stream, the capture and the memory query) must
initialise the GPU before libptamd.so does."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cuda-path-tracer_amd", "python"))


def main():
    import torch

    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda:0")
    import ptamd as pt
    from ptamd import scenes

    desc = scenes.teapot_scene(res=(640, 360))
    g, cam = pt.scene_from_description(desc)
    n_pixels, n_split, s1 = cam.res_x * cam.res_y, 4, 256
    ref = pt.DeviceBuffer(0, n_split * n_pixels * 12)
    ref.zero()
    g.render_accumulate(cam, 9, 0, s1, n_split, ref.ptr)
    want = ref.download((n_split, cam.res_y, cam.res_x, 3))
    ref.free()
    assert want.sum() > 0
    # 4 x 8 slice images do not fit the scratch: this call renders in groups of slices
    assert 32 * n_pixels * 12 > pt.lib().pt_noalloc_max_partial_bytes()
    split = torch.zeros((n_split, cam.res_y, cam.res_x, 3), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]
    g.render_accumulate_noalloc(cam, 9, 0, s1, n_split, split.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info(0)[0] == free0, "device memory in use changed over the call"
    assert np.array_equal(split.cpu().numpy().view(np.uint32), want.view(np.uint32))
    stream = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    split.zero_()
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=stream):
        g.render_accumulate_noalloc(cam, 9, 0, s1, n_split, split.data_ptr(), torch.cuda.current_stream().cuda_stream)
    for _ in range(2):
        split.zero_()
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        got = split.cpu().numpy()
        same = np.array_equal(got.view(np.uint32), want.view(np.uint32))
        if not same:
            d = np.abs(got - want)
            print("replay differs: nonzero got", int((got != 0).sum()), "of", got.size, "max diff", float(d.max()), "pixels differing per split",
                  [(int((d[z].max(axis=-1) > 0).sum())) for z in range(n_split)], flush=True)
        assert same
    del graph
    g.close()
    print("noalloc graph ok")


if __name__ == "__main__":
    main()
