"""Depth pre-filter ("next" row f3): restatement of cv::cuda::bilateralFilter(depth, depth, -1, 0.03, 4.5)
(core/src/supersurfel_fusion.cu:180).  OpenCV is a third-party dependency that is not vendored by the
reference and not installed here, so the algorithm is checked against an independent numpy evaluation
of its published definition (parity unpinned) -- and HIP against the oracle bit for bit."""
import ctypes as C

import numpy as np
import pytest

import util
from supersurfel_fusion_amd import binding


def numpy_bilateral(d, sigma_color=0.03, sigma_space=4.5):
    H, W = d.shape
    r = int(round(sigma_space * 1.5))
    pad = np.pad(d.astype(np.float64), r, mode="reflect")
    num = np.zeros((H, W)); den = np.zeros((H, W))
    for dy in range(-r, r + 1):
        for dx in range(-r, r + 1):
            if dx * dx + dy * dy > r * r:
                continue
            v = pad[r + dy:r + dy + H, r + dx:r + dx + W]
            w = np.exp(-(dx * dx + dy * dy) / (2 * sigma_space ** 2) - (v - d) ** 2 / (2 * sigma_color ** 2))
            num += w * v; den += w
    return num / den


def test_specified_exp_is_accurate(oracle_lib):
    """exp_neg: |rel err| <= 1 ulp of float against numpy's double exp over the filter's argument range"""
    f = binding.Fusion(oracle_lib, util.make_cfg(oracle_lib, 64, 48, cell_size=16, nb_supersurfels_max=64))
    # a 1-pixel-wide response: filter a step image and compare a few taps instead of exposing exp directly
    d = np.full((48, 64), 1.0, np.float32); d[:, 32:] = 1.02
    out = f.bilateral_filter(d)
    ref = numpy_bilateral(d)
    assert np.abs(out - ref).max() < 2e-6


def test_bilateral_matches_definition(oracle_lib):
    rng = np.random.default_rng(2)
    rgb, depth = util.frame(0, 160, 128, noise=True, holes=0.05)
    f = binding.Fusion(oracle_lib, util.make_cfg(oracle_lib, 160, 128, nb_supersurfels_max=4096))
    out = f.bilateral_filter(depth)
    ref = numpy_bilateral(depth)
    assert np.abs(out - ref).max() < 5e-6
    # edge preserving: holes (0) stay (near) 0 next to metre-scale depths, noise on flat parts shrinks
    assert np.abs(out[depth == 0]).max() < 1e-3
    flat = (depth > 0) & (np.abs(out - depth) < 0.02)
    assert flat.mean() > 0.9


def test_prefilter_inside_process_frame(oracle_lib):
    rgb, depth = util.frame(0, 160, 128, noise=True)
    fa = binding.Fusion(oracle_lib, util.make_cfg(oracle_lib, 160, 128, nb_supersurfels_max=4096, depth_prefilter=1))
    fb = binding.Fusion(oracle_lib, util.make_cfg(oracle_lib, 160, 128, nb_supersurfels_max=4096))
    fa.process_frame(rgb, depth)
    fb.process_frame(rgb, fb.bilateral_filter(depth))
    util.compare_state(fa, fb)


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(160, 128), (150, 100), (640, 480)])
def test_prefilter_bit_exact_on_gpu(size, oracle_lib, product_lib):
    W, H = size
    rgb, depth = util.frame(1, W, H, noise=True, holes=0.05)
    fo = binding.Fusion(oracle_lib, util.make_cfg(oracle_lib, W, H, nb_supersurfels_max=8192, depth_prefilter=1))
    fh = binding.Fusion(product_lib, util.make_cfg(product_lib, W, H, nb_supersurfels_max=8192, depth_prefilter=1))
    util.assert_same_bits(fo.bilateral_filter(depth), fh.bilateral_filter(depth), "bilateral filter")
    if W <= 160:
        for k in range(2):
            rgb, depth = util.frame(k, W, H, noise=True, holes=0.05)
            util.same_result(fo.process_frame(rgb, depth), fh.process_frame(rgb, depth))
        util.compare_state(fo, fh)
