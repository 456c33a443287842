"""Per-element arithmetic: known answers from the reference's own math headers and bitwise
oracle-vs-product agreement of the host-evaluated kernel arithmetic (CPU, no GPU needed).

Known answers: the reference headers vector_math.cuh / matrix_math.cuh were
compiled verbatim on the host during the survey: rgbToLab(120,30,200) = (36.160343, 65.641357,
-69.750305) and the round trip labToRgb(rgbToLab(.)) = (120.002274, 30.001722, 200.000336).
Those used libm powf/cbrtf; this build replaces them by a specified IEEE sequence, so the
comparison is to 2e-4 (Lab units / 8-bit colour units)."""
import ctypes as C

import numpy as np
import pytest


def fptr(a):
    return a.ctypes.data_as(C.c_void_p)


def call3(L, fn, x, nout=3):
    x = np.ascontiguousarray(x, np.float32); o = np.zeros(nout, np.float32)
    getattr(L, fn).argtypes = [C.c_void_p, C.c_void_p]
    getattr(L, fn)(fptr(x), fptr(o))
    return o


@pytest.mark.parametrize("which", ["oracle", "product"])
def test_lab_known_answer_from_reference_headers(which, oracle_lib, product_lib):
    L = (oracle_lib if which == "oracle" else product_lib).lib
    lab = call3(L, "ssf_dbg_rgb_to_lab", [120, 30, 200])
    assert np.allclose(lab, [36.160343, 65.641357, -69.750305], atol=2e-4)
    rgb = call3(L, "ssf_dbg_lab_to_rgb", lab)
    assert np.allclose(rgb, [120.002274, 30.001722, 200.000336], atol=2e-4)
    # CIE anchors: white -> L=100, black -> 0
    assert np.allclose(call3(L, "ssf_dbg_rgb_to_lab", [255, 255, 255]), [100, 0, 0], atol=2e-2)
    assert np.allclose(call3(L, "ssf_dbg_rgb_to_lab", [0, 0, 0]), [0, 0, 0], atol=1e-5)


def test_lab_matches_double_precision_formula(oracle_lib):
    """The specified Newton roots agree with an fp64 evaluation of the same colour formulas."""
    rng = np.random.default_rng(7)
    for c in rng.uniform(0, 255, (200, 3)):
        lab = call3(oracle_lib.lib, "ssf_dbg_rgb_to_lab", c)
        v = np.float32(c).astype(np.float64) / 255.0
        lin = np.where(v > 0.04045, ((v + 0.055) / 1.055) ** 2.4, v / 12.92)
        x = (lin @ [0.4124, 0.3575, 0.1805]) / 0.95047; y = lin @ [0.2126, 0.7152, 0.0722]
        z = (lin @ [0.0193, 0.1192, 0.9505]) / 1.08883
        f = lambda t: np.cbrt(t) if t > 0.008856 else 7.787 * t + 16.0 / 116.0
        ref = [116 * f(y) - 16, 500 * (f(x) - f(y)), 200 * (f(y) - f(z))]
        assert np.allclose(lab, ref, atol=2e-4)


def test_kernel_arithmetic_bitwise_oracle_vs_product(oracle_lib, product_lib):
    rng = np.random.default_rng(11)
    Lo, Lp = oracle_lib.lib, product_lib.lib
    for i in range(400):
        rgb = rng.uniform(-5, 260, 3)
        a, b = call3(Lo, "ssf_dbg_rgb_to_lab", rgb), call3(Lp, "ssf_dbg_rgb_to_lab", rgb)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) or (np.isnan(a).any() and np.isnan(b).any())
        lab = rng.uniform([0, -120, -120], [100, 120, 120])
        a, b = call3(Lo, "ssf_dbg_lab_to_rgb", lab), call3(Lp, "ssf_dbg_lab_to_rgb", lab)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        M = rng.standard_normal((3, 3)) * rng.uniform(1e-3, 1.0)
        Cm = M @ M.T + np.eye(3) * 1e-7
        cov = np.float32([Cm[0, 0], Cm[0, 1], Cm[0, 2], Cm[1, 1], Cm[1, 2], Cm[2, 2]])
        a, b = call3(Lo, "ssf_dbg_sym_inverse", cov, 6), call3(Lp, "ssf_dbg_sym_inverse", cov, 6)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        for L, store in ((Lo, []), (Lp, [])):
            pass
        vo, wo, vp, wp = np.zeros(9, np.float32), np.zeros(3, np.float32), np.zeros(9, np.float32), np.zeros(3, np.float32)
        for L, v, w in ((Lo, vo, wo), (Lp, vp, wp)):
            L.ssf_dbg_principal_frame.argtypes = [C.c_void_p] * 3
            L.ssf_dbg_principal_frame(fptr(cov), fptr(v), fptr(w))
        assert np.array_equal(vo.view(np.uint32), vp.view(np.uint32)) and np.array_equal(wo.view(np.uint32), wp.view(np.uint32))
        rows = np.float32(rng.uniform(-50, 50, 12))
        a, b = call3(Lo, "ssf_dbg_plane_solve", rows), call3(Lp, "ssf_dbg_plane_solve", rows)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_principal_frame_recovers_known_axes(oracle_lib):
    """eigenDecomposition by repeated squaring: major axis, normal and eigenvalues of a known
    covariance (rows = major, normal x major, normal; supersurfel_fusion_kernels.cu:48-111)."""
    R = np.array([[0.6, 0.8, 0.0], [-0.8, 0.6, 0.0], [0.0, 0.0, 1.0]])   # rows are the axes
    lam = np.array([4e-4, 1e-4, 4e-6])
    Cm = R.T @ np.diag(lam) @ R
    cov = np.float32([Cm[0, 0], Cm[0, 1], Cm[0, 2], Cm[1, 1], Cm[1, 2], Cm[2, 2]])
    v, w = np.zeros(9, np.float32), np.zeros(3, np.float32)
    oracle_lib.lib.ssf_dbg_principal_frame.argtypes = [C.c_void_p] * 3
    oracle_lib.lib.ssf_dbg_principal_frame(fptr(cov), fptr(v), fptr(w))
    v = v.reshape(3, 3)
    assert abs(abs(v[0] @ R[0]) - 1) < 1e-4 and abs(abs(v[2] @ R[2]) - 1) < 1e-4
    assert np.allclose(w, lam, rtol=2e-3)


def test_sym_inverse_threshold(oracle_lib):
    """|det| > 1e-9 gate of inverse(Cov3) (matrix_math.cuh:51): near-singular input is rejected."""
    L = oracle_lib.lib
    L.ssf_dbg_sym_inverse.argtypes = [C.c_void_p, C.c_void_p]
    ok = np.float32([1e-2, 0, 0, 1e-2, 0, 1e-2]); bad = np.float32([1e-4, 0, 0, 1e-4, 0, 1e-4]); o = np.zeros(6, np.float32)
    assert L.ssf_dbg_sym_inverse(fptr(ok), fptr(o)) == 1 and np.allclose(o[[0, 3, 5]], 100.0, rtol=1e-6)
    assert L.ssf_dbg_sym_inverse(fptr(bad), fptr(o)) == 0
