"""The allocation-free render entry point (pt_render_accumulate_noalloc) against pt_render_accumulate.

Both run the same kernels with the same slice plan, so the split buffers must hold identical bits.
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


def test_noalloc_rejects_sizes_beyond_the_scratch(pt, scenes):
    desc = scenes.cornellbox_scene(res=(1920, 1080))
    g, cam = pt.scene_from_description(desc)
    n_pixels = cam.res_x * cam.res_y
    limit = pt.lib().pt_noalloc_max_partial_bytes()
    assert 2 * n_pixels * 12 <= limit < 3 * n_pixels * 12
    split = pt.DeviceBuffer(0, 4 * n_pixels * 12)
    split.zero()
    with pytest.raises(pt.PtError, match="exceed the fixed scratch"):
        g.render_accumulate_noalloc(cam, 1, 0, 64, 4, split.ptr)      # 4 x 2 slices
    assert not split.download((16,)).any()
    g.render_accumulate_noalloc(cam, 1, 0, 16, 1, split.ptr)          # 2 slices: fits
    ref = pt.DeviceBuffer(0, n_pixels * 12)
    ref.zero()
    g.render_accumulate(cam, 1, 0, 16, 1, ref.ptr)
    assert np.array_equal(split.download((n_pixels, 3)).view(np.uint32), ref.download((n_pixels, 3)).view(np.uint32))
    g.close()
