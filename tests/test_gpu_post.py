"""GPU tests of the post-processing path against the CPU oracle.  Floating point, different
summation order and the device powf: tolerance 1e-5 absolute on filter outputs of O(1) images
(relative 1e-5 for HDR inputs), 2e-5 on the full tone-mapped chain."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _img(rng, h, w, scale=1.0):
    return (rng.random((h, w, 3), dtype=np.float32) * scale).astype(np.float32)


@pytest.mark.parametrize("ksize,sigma", [(0, 1.0), (1, 0.8), (2, 0.7), (7, 3.0), (32, 10.0)])
def test_normal_filter(pt, oracle, ksize, sigma):
    rng = np.random.default_rng(ksize)
    for (h, w) in ((45, 70), (64, 96), (8, 8)):
        img = _img(rng, h, w)
        taps = oracle.make_gaussian(ksize, sigma)
        got = pt.apply_filter_host(img, taps, bilateral=False)
        want = oracle.apply_filter(img, taps, bilateral=False, n_threads=8)
        assert np.abs(got - want).max() <= 1e-5


def test_bloom_sized_gaussian_runs_as_two_passes(pt, oracle):
    """65x65 sigma-10 table (an outer product: the library filters it as a row and a column pass) on an
    image narrower than the window in one direction, sparse HDR input like the thresholded bloom layer."""
    rng = np.random.default_rng(5)
    taps = oracle.make_gaussian(32, 10.0)
    for (h, w) in ((70, 150), (40, 300), (9, 33)):
        img = _img(rng, h, w, scale=40.0)
        img[img < 36] = 0
        got = pt.apply_filter_host(img, taps)
        want = oracle.apply_filter(img, taps, n_threads=8)
        # 4225 fp32 additions per pixel in a different order: a few 1e-6 relative either way
        assert np.abs(got - want).max() <= 1e-5 * float(want.max()), (h, w)


def test_taps_that_are_not_an_outer_product(pt, oracle):
    rng = np.random.default_rng(6)
    img = _img(rng, 37, 53)
    taps = rng.random((7, 7), dtype=np.float32) + 0.05
    got = pt.apply_filter_host(img, taps)
    want = oracle.apply_filter(img, taps, n_threads=8)
    assert np.abs(got - want).max() <= 1e-5
    plus = np.zeros((5, 5), np.float32)      # zero taps: a "plus" shape has rank 2
    plus[2, :] = 1.0
    plus[:, 2] = 1.0
    got = pt.apply_filter_host(img, plus)
    want = oracle.apply_filter(img, plus, n_threads=8)
    assert np.abs(got - want).max() <= 1e-5


def test_mean_filter_and_hdr_input(pt, oracle):
    rng = np.random.default_rng(3)
    img = _img(rng, 50, 61, scale=40.0)
    img[img < 30] = 0          # sparse bright pixels, like the thresholded bloom layer
    taps = oracle.make_mean(3)
    got = pt.apply_filter_host(img, taps)
    want = oracle.apply_filter(img, taps, n_threads=8)
    assert np.abs(got - want).max() <= 1e-5 * max(1.0, float(want.max()))


@pytest.mark.parametrize("ksize,sigma", [(2, 0.7), (4, 2.0)])
def test_bilateral_filter(pt, oracle, ksize, sigma):
    rng = np.random.default_rng(11)
    img = _img(rng, 57, 83)
    img[20:30, 30:50] += 0.5   # an edge the range kernel must preserve
    img = np.clip(img, 0, 1).astype(np.float32)
    taps = oracle.make_gaussian(ksize, sigma)
    got = pt.apply_filter_host(img, taps, bilateral=True)
    want = oracle.apply_filter(img, taps, bilateral=True, n_threads=8)
    assert np.abs(got - want).max() <= 1e-5


def test_elementwise_ops(pt, oracle):
    rng = np.random.default_rng(2)
    img = _img(rng, 33, 47, scale=6.0)
    layer = _img(rng, 33, 47, scale=2.0)
    n = 33 * 47
    L = pt.lib()
    a = pt.DeviceBuffer(0, img.nbytes)
    b = pt.DeviceBuffer(0, img.nbytes)
    a.upload(img)
    pt.check(L.pt_normalize_image(0, a.ptr, n, 48, None))
    assert np.array_equal(a.download(img.shape), oracle.normalize_image(img, 48))
    a.upload(img)
    pt.check(L.pt_apply_threshold(0, a.ptr, b.ptr, n, 3.0, None))
    assert np.array_equal(b.download(img.shape), oracle.apply_threshold(img, 3.0))
    a.upload(img)
    b.upload(layer)
    pt.check(L.pt_image_add(0, a.ptr, b.ptr, n, 0.5, None))
    assert np.array_equal(a.download(img.shape), oracle.image_add(img, layer, 0.5))
    a.upload(img)
    pt.check(L.pt_apply_aces(0, a.ptr, n, None))
    assert np.abs(a.download(img.shape) - oracle.apply_aces(img)).max() <= 1e-6
    unit = np.clip(img / 6.0, 0, 1).astype(np.float32)
    a.upload(unit)
    pt.check(L.pt_gamma_correct(0, a.ptr, n, 2.2, None))
    assert np.abs(a.download(img.shape) - oracle.gamma_correct(unit, 2.2)).max() <= 1e-6
    split = _img(rng, 4 * 33, 47)
    s = pt.DeviceBuffer(0, split.nbytes)
    s.upload(split)
    pt.check(L.pt_compound_buffers(0, s.ptr, a.ptr, n, 4, None))
    assert np.array_equal(a.download((n, 3)), oracle.compound_buffers(split, 4))


def test_full_post_chain(pt, oracle):
    rng = np.random.default_rng(21)
    h, w, spp = 72, 128, 64
    img = _img(rng, h, w, scale=1.5)
    img[10:14, 20:24] = 40.0     # a few pixels above the bloom threshold
    img[50:52, 100:103] = 15.0
    summed = (img * spp).astype(np.float32)
    got = pt.post_chain_host(summed, spp)
    want = oracle.post_chain(summed, spp, (w, h))
    assert got.min() >= 0 and got.max() <= 1.0 + 1e-6
    assert np.abs(got - want).max() <= 2e-5
    assert np.array_equal(oracle.to_ppm_bytes(got) // 2, oracle.to_ppm_bytes(want) // 2) or \
        (np.abs(oracle.to_ppm_bytes(got).astype(int) - oracle.to_ppm_bytes(want).astype(int)).max() <= 1)
