"""The allocation-free render entry point (pt_render_accumulate_noalloc) against pt_render_accumulate.

Both run the same kernels on the same slice plan and add the slices up in the same order (at once, or a few slices
per launch when the plan's images do not fit the scratch), so the split buffers must hold identical bits.
Two calls in a row, the second with a smaller input: the fixed scratch array keeps the first call's
contents (queue cursors, slice partials), and nothing of them may leak into the second result."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(pt, g, cam, seed, s0, s1, n_split):
    n_pixels = cam.res_x * cam.res_y
    out = []
    for noalloc in (False, True):
        split = pt.DeviceBuffer(0, n_split * n_pixels * 12)
        split.zero()
        if noalloc:
            g.render_accumulate_noalloc(cam, seed, s0, s1, n_split, split.ptr)
        else:
            g.render_accumulate(cam, seed, s0, s1, n_split, split.ptr)
        out.append(split.download((n_split, cam.res_y, cam.res_x, 3)))
        split.free()
    return out


def test_noalloc_matches_allocating_entry_point_twice(pt, scenes):
    big = scenes.teapot_scene(res=(160, 96))
    g, cam = pt.scene_from_description(big)
    a, b = _both(pt, g, cam, 1337, 0, 48, 4)             # 4 split buffers x 2 slices of 6 samples
    assert a.sum() > 0 and np.isfinite(a).all()
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    # second call, smaller input: another scene (textured, HDRI, heavy kernel build), fewer pixels, fewer samples
    small = scenes.hcandersen_scene(res=(72, 40), n_tris=6000, tex_res=128, hdri_res=(128, 64))
    g2, cam2 = pt.scene_from_description(small)
    g2.set_launch("heavy", None)
    a2, b2 = _both(pt, g2, cam2, 7, 3, 12, 2)
    assert a2.sum() > 0
    assert np.array_equal(a2.view(np.uint32), b2.view(np.uint32))
    # and the first scene again through the scratch the second call left behind
    a3, b3 = _both(pt, g, cam, 1337, 0, 48, 4)
    assert np.array_equal(a3.view(np.uint32), a.view(np.uint32))
    assert np.array_equal(b3.view(np.uint32), a.view(np.uint32))
    g.close()
    g2.close()


def test_noalloc_renders_the_headline_frame_in_slice_groups(pt, scenes):
    """1920x1080 with the reference's four split buffers: the plan's eight slice images (199 MB) do not fit the 64 MiB
    scratch, so the call renders the slices one at a time, keeping the running sum of a split buffer's slices in the
    scratch -- the same additions in the same order as ONE pt_render_accumulate call: identical bits."""
    desc = scenes.cornellbox_scene(res=(1920, 1080))
    g, cam = pt.scene_from_description(desc)
    n_pixels = cam.res_x * cam.res_y
    limit = pt.lib().pt_noalloc_max_partial_bytes()
    assert 2 * n_pixels * 12 <= limit < 3 * n_pixels * 12
    a, b = _both(pt, g, cam, 1, 0, 64, 4)                 # 4 split buffers x 2 slices of 8 samples, in 8 launches of one slice
    assert a.sum() > 0 and np.isfinite(a).all()
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    a, b = _both(pt, g, cam, 1, 5, 21, 1)                 # 2 slices: fit together, one launch
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    g.close()


def test_noalloc_rejects_frames_beyond_the_scratch(pt, scenes):
    desc = scenes.cornellbox_scene(res=(2048, 1400))      # 34.4 MB per image: two do not fit
    g, cam = pt.scene_from_description(desc)
    n_pixels = cam.res_x * cam.res_y
    assert pt.lib().pt_noalloc_max_partial_bytes() < 2 * n_pixels * 12
    split = pt.DeviceBuffer(0, 2 * n_pixels * 12)
    split.zero()
    with pytest.raises(pt.PtError, match="needs two images"):
        g.render_accumulate_noalloc(cam, 1, 0, 64, 2, split.ptr)
    assert not split.download((16,)).any()
    g.render_accumulate_noalloc(cam, 1, 0, 8, 1, split.ptr)           # one slice: fits
    assert split.download((n_pixels, 3)).sum() > 0
    split.free()
    g.close()


def test_noalloc_allocates_nothing_and_runs_under_graph_capture():
    """The property the entry point exists for: device memory in use is the same before and after a call, and the call
    is legal while a HIP graph is being captured (grouped launches included); two replays of the graph give the bits of
    pt_render_accumulate each time.  In a process of its own (tests/noalloc_graph_worker.py): torch, whose streams and
    graph capture the check uses, has to initialise the GPU before the library does."""
    import os
    import subprocess
    import sys

    from conftest import REPO

    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "noalloc_graph_worker.py")], capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
    assert "noalloc graph ok" in r.stdout
