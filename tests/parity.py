"""Shared criteria for comparing a HIP render with the oracle's render of the same samples.

Both sides draw the same Philox numbers per (pixel, sample, bounce) and run the same geometry bit for bit, so
a path differs only where a last-bit difference of a hardware transcendental (sin / cos / exp2 / log2 / rcp /
rsq in shading; libm in the oracle) flips a discrete decision further down the path: measured 0.6e-5 to
1.5e-5 of the paths (profiles/diag/path_diff.py), each by a whole bounce -- e.g. one more hit of a light of
emission 10 -- so a handful of such paths can dominate an RMSE.  The criteria therefore are:
  1. per pixel: all but at most 1 % of the pixels agree within 2e-4 * (image RMS + |pixel|)  (at 32-64 spp
     the expected share of pixels holding a flipped path is 0.05-0.1 %);
  2. whole image: RMSE(GPU, oracle) <= 25 % of the Monte-Carlo noise RMSE(oracle @ spp, oracle @ 16 x spp),
     and the GPU image is as far from the converged image as the oracle's is (within 15 %);
  3. work counters within 0.5 %, path count exact.
"""
import numpy as np


def rmse(a, b):
    return float(np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)))


def assert_pixels_agree(img_g, img_c, max_flipped=0.01, tol=2e-4):
    """img_g, img_c: normalised radiance (H, W, 3).  Returns the share of pixels that differ."""
    assert np.isfinite(img_g).all()
    g, c = img_g.astype(np.float64), img_c.astype(np.float64)
    rms = float(np.sqrt(np.mean(c ** 2)))
    assert rms > 0
    diff = np.abs(g - c).max(axis=-1)
    flipped = diff > tol * (rms + np.abs(c).max(axis=-1))
    share = float(flipped.mean())
    assert share <= max_flipped, (share, float(diff.max()), rms)
    return share


def assert_render_parity(img_g, img_c, conv, cnt_g=None, cnt_c=None, n_paths=None):
    """img_g / img_c: normalised GPU / oracle radiance of the same samples; conv: the oracle's normalised render
    of 16 x as many other samples."""
    assert_pixels_agree(img_g, img_c)
    err, noise_c, noise_g = rmse(img_g, img_c), rmse(img_c, conv), rmse(img_g, conv)
    assert err <= 0.25 * noise_c, (err, noise_c)
    assert abs(noise_g - noise_c) <= 0.15 * noise_c, (noise_g, noise_c)
    if cnt_g is not None:
        if n_paths is not None:
            assert cnt_g["paths"] == cnt_c["paths"] == n_paths
        for k in ("segments", "tri_tests", "node_visits", "env_lookups", "tex_fetches"):
            assert abs(cnt_g[k] - cnt_c[k]) <= 0.005 * max(cnt_c[k], 1) + 2, (k, cnt_g[k], cnt_c[k])
