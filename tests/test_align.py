"""Loop-closure registration: ssf_align = DenseRegistration::align
(dense_registration.cu:52-243) of a keyframe's supersurfels against the current frame, and the fern
encoding (ferns_kernels.cu:48-70).  CPU: behaviour of the oracle restatement; GPU: HIP == oracle bit for bit."""
import numpy as np
import pytest

import util
from supersurfel_fusion_amd import binding, synthetic


def keyframe_and_current(lib, W=320, H=240, k_key=0, k_cur=3, **cfg):
    """Frame supersurfels of orbit frame k_key (the 'fern keyframe') and a handle whose current frame is k_cur."""
    fk = binding.Fusion(lib, util.make_cfg(lib, W, H, **cfg))
    fk.process_frame(*util.frame(k_key, W, H))
    key = fk.get_frame()
    valid = key["confidences"] > 0
    src = {name: key[name][valid] for name in ("positions", "colors", "orientations")}
    fc = binding.Fusion(lib, util.make_cfg(lib, W, H, **cfg))
    fc.process_frame(*util.frame(k_cur, W, H))
    # ground truth: keyframe camera -> current camera
    Rk, tk = synthetic.orbit_pose(k_key); Rc, tc = synthetic.orbit_pose(k_cur)
    R = Rc.T @ Rk; t = Rc.T @ (tk - tc)
    return src, fc, synthetic.pose12(R, t)


def perturbed(gt12, seed=5):
    """A PnP-quality prior (1 degree, 2 cm off the ground truth): align works with a 0.1 m / 20 Lab gate."""
    rng = np.random.default_rng(seed)
    R = gt12[:9].reshape(3, 3).astype(np.float64); t = gt12[9:].astype(np.float64)
    ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
    a = np.deg2rad(1.0)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    dR = np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K
    return np.concatenate([(dR @ R).reshape(9), t + rng.normal(size=3) * 0.012]).astype(np.float32)


def test_align_recovers_the_relative_pose(oracle_lib):
    src, fc, gt = keyframe_and_current(oracle_lib, 320, 240)
    init = perturbed(gt)
    out = fc.align(src, init)
    assert out["valid"] and out["iters"] == 10 and out["pairs"] > 100
    # (R, t) returned = inverse of the accumulated increment: composing it with the prior must approach the truth
    Rr = out["rel_pose"][:9].reshape(3, 3).astype(np.float64); tr = out["rel_pose"][9:].astype(np.float64)
    Ri = init[:9].reshape(3, 3).astype(np.float64); ti = init[9:].astype(np.float64)
    R_est = Rr.T @ Ri; t_est = Rr.T @ (ti - tr)          # R_inc * R_init, R_inc * t_init + t_inc
    Rg = gt[:9].reshape(3, 3); tg = gt[9:]
    err_before = np.linalg.norm(ti - tg); err_after = np.linalg.norm(t_est - tg)
    ang_after = np.degrees(np.arccos(np.clip((np.trace(R_est @ Rg.T) - 1) / 2, -1, 1)))
    assert err_after < 0.6 * err_before and ang_after < 0.5


def test_align_rejects_when_there_are_too_few_pairs(oracle_lib):
    src, fc, gt = keyframe_and_current(oracle_lib, 320, 240)
    far = gt.copy(); far[9:] += 2.0                      # 2 m off: no pair passes the 0.1 m gate
    out = fc.align(src, far)
    assert not out["valid"] and out["iters"] == 1 and out["pairs"] < 100
    assert np.array_equal(out["rel_pose"], np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], np.float32))


def test_fern_codes_known_answer(oracle_lib):
    W, H = 64, 48
    f = binding.Fusion(oracle_lib, util.make_cfg(oracle_lib, 160, 128))
    rgb = np.zeros((H, W, 3), np.uint8); rgb[..., 0] = 200; rgb[..., 1] = 10; rgb[..., 2] = 100
    depth = np.full((H, W), 2.0, np.float32)
    pos = np.array([[3, 4], [63, 47], [1000, 1000]], np.uint32)        # the last one clamps to the border texel
    frgb = np.array([[100, 100, 100], [250, 5, 100], [0, 0, 0]], np.uint8)
    fd = np.array([1.0, 3.0, 2.0], np.float32)
    assert f.fern_codes(rgb, depth, pos, frgb, fd).tolist() == [1 | 8, 2, 1 | 2 | 4]


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(320, 240), (640, 480)])
def test_align_hip_equals_oracle(size, oracle_lib, product_lib):
    W, H = size
    so, fo, gt = keyframe_and_current(oracle_lib, W, H)
    sh, fh, _ = keyframe_and_current(product_lib, W, H)
    for name in so:
        util.assert_same_bits(so[name], sh[name], "keyframe " + name)
    for init in (perturbed(gt), perturbed(gt, seed=9), None):
        a, b = fo.align(so, init), fh.align(sh, init)
        assert (a["valid"], a["iters"], a["pairs"]) == (b["valid"], b["iters"], b["pairs"]), (a, b)
        util.assert_same_bits(a["rel_pose"], b["rel_pose"], "align rel_pose")
    conf = np.ones(len(so["positions"]), np.float32); conf[::3] = -1.0   # with a confidence array
    a, b = fo.align(dict(so, confidences=conf), perturbed(gt)), fh.align(dict(sh, confidences=conf), perturbed(gt))
    assert (a["valid"], a["iters"], a["pairs"]) == (b["valid"], b["iters"], b["pairs"])
    util.assert_same_bits(a["rel_pose"], b["rel_pose"], "align rel_pose (confidences)")


@pytest.mark.gpu
def test_fern_codes_hip_equals_oracle(oracle_lib, product_lib):
    rng = np.random.default_rng(3)
    W, H, n = 160, 120, 500
    rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8); depth = rng.uniform(0, 5, (H, W)).astype(np.float32)
    pos = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], 1).astype(np.uint32)
    frgb = rng.integers(0, 256, (n, 3), dtype=np.uint8); fd = rng.uniform(0, 5, n).astype(np.float32)
    fo = binding.Fusion(oracle_lib, util.make_cfg(oracle_lib, 160, 128))
    fh = binding.Fusion(product_lib, util.make_cfg(product_lib, 160, 128))
    assert np.array_equal(fo.fern_codes(rgb, depth, pos, frgb, fd), fh.fern_codes(rgb, depth, pos, frgb, fd))
