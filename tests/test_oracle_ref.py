"""Pins the oracle's REF_SSE mode against the REFERENCE'S OWN code.

oracle/_ref/libdvo_ref.so is built from the reference's translation units dense_tracking_impl.cpp, core/math_sse.cpp and
core/intrinsic_matrix.cpp, compiled unmodified where they lie under /root/reference against stand-in headers for Eigen / OpenCV /
boost (oracle/shim/, oracle/ref_bridge.cpp, oracle/Makefile).  Each pass of an iteration is run through the reference function
and through the oracle's restatement on the same arrays; outputs must be BIT-IDENTICAL."""
import ctypes as C

import numpy as np
import pytest

import common as cm
from oracle import pyoracle as po

ref = po.ref_lib()
pytestmark = pytest.mark.skipif(ref is None, reason="oracle/_ref not built (reference tree absent)")


def fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def f32(*v):
    return np.ascontiguousarray(v, dtype=np.float32)


def level_arrays(pyr, level):
    """accel [h,w,8] and the selected points [n,12] of one pyramid level, built like the reference does
    (rgbd_image.cpp:534-543, 245-262; point_selection.cpp:89-152)."""
    planes = [pyr.plane(level, k)[0] for k in range(6)]
    K = pyr.plane(level, 0)[1]
    h, w = planes[0].shape
    accel = np.zeros((h, w, 8), np.float32)
    for k in range(6):
        accel[..., k] = planes[k]
    n_sel, mask = pyr.select(level)
    ys, xs = np.nonzero(mask)
    z = planes[1][ys, xs]
    pts = np.zeros((len(ys), 12), np.float32)
    pts[:, 0] = ((xs.astype(np.float32) - K[2]) / K[0]) * z
    pts[:, 1] = ((ys.astype(np.float32) - K[3]) / K[1]) * z
    pts[:, 2] = z
    pts[:, 3] = 1.0
    pts[:, 4:12] = accel[ys, xs]
    assert len(ys) == n_sel
    return np.ascontiguousarray(accel), np.ascontiguousarray(pts), K, w, h


def run_residuals(which, pts, accel, w, h, K, T34):
    n = len(pts)
    out_p = np.full((n + 2, 12), np.nan, np.float32)
    out_r = np.full((n + 2, 2), np.nan, np.float32)
    T = np.ascontiguousarray(T34, np.float32).reshape(-1)
    if which == "ref":
        wr, wc = np.zeros(8, np.float32), np.zeros(8, np.float32)
        po.lib().oracle_pass_weight_vectors(fp(K), fp(wr), fp(wc))          # dense_tracking.cpp:215-220, same floats on both sides
        n_out = ref.ref_compute_residuals(1, n, fp(pts), fp(accel), w, h, fp(K), fp(T), fp(wr), fp(wc), fp(out_p), fp(out_r))
    else:
        n_out = po.lib().oracle_pass_residuals(po.REF_SSE, n, fp(pts), fp(accel), w, h, fp(K), fp(T), fp(out_p), fp(out_r))
    return n_out, out_p[:n_out], out_r[:n_out]


@pytest.mark.parametrize("seed,w,h,level", [(11, 320, 240, 0), (12, 320, 240, 2), (13, 160, 120, 1), (14, 100, 76, 0)])
def test_residual_pass_is_the_references(seed, w, h, level):
    pair = cm.synth(seed, w, h)
    oref, ocur = cm.oracle_pyramids(pair, 3)
    _, pts, K, lw, lh = level_arrays(oref, level)
    accel, _, _, _, _ = level_arrays(ocur, level)
    rng = np.random.default_rng(seed)
    for k, xi in enumerate([np.zeros(6), -pair["xi_true"], rng.uniform(-0.05, 0.05, 6), rng.uniform(-0.3, 0.3, 6)]):
        T34 = po.se3_exp(xi)[:3]
        p = pts if k % 2 == 0 else pts[:-1 if len(pts) % 2 == 0 else len(pts)]     # even and odd point counts (Q3)
        n_r, pr, rr = run_residuals("ref", p, accel, lw, lh, K, T34)
        n_o, po_, ro = run_residuals("oracle", p, accel, lw, lh, K, T34)
        assert n_r == n_o and (k >= 2 or n_r > 0.1 * len(p)), (n_r, n_o, len(p))     # small motions keep most points
        assert pr.tobytes() == po_.tobytes()          # compacted points with all 8 combined channels
        assert rr.tobytes() == ro.tobytes()           # residual pairs


@pytest.mark.parametrize("n", [4, 7, 1000, 1003, 20001])
def test_weights_scale_and_loglik_passes_are_the_references(n):
    rng = np.random.default_rng(n)
    res = np.ascontiguousarray(rng.normal(size=(n, 2)) * [0.02, 0.01], np.float32)
    res[rng.integers(0, n, max(1, n // 50))] *= 30.0          # outliers
    P = f32(2500.0, -300.0, -300.0, 9000.0)
    zero = f32(0, 0)
    w_ref, w_ora = np.zeros(n, np.float32), np.zeros(n, np.float32)
    ref.ref_compute_weights(1, n, fp(res), fp(zero), fp(P), fp(w_ref))
    po.lib().oracle_pass_weights(po.REF_SSE, n, fp(res), fp(P), fp(w_ora))
    assert w_ref.tobytes() == w_ora.tobytes()
    for weights in (w_ref, np.ones(n, np.float32)):           # first iteration of a level: unit weights
        S = np.zeros(4, np.float32)
        Cc = np.zeros(3, np.float32)
        ref.ref_compute_scale(1, n, fp(res), fp(weights), fp(zero), fp(S))
        po.lib().oracle_pass_scale(po.REF_SSE, n, fp(res), fp(weights), fp(Cc))
        assert S[[0, 1, 3]].tobytes() == Cc.tobytes() and S[1].tobytes() == S[2].tobytes()
    ll_ref = ref.ref_loglik(n, fp(res), fp(w_ref), fp(zero), fp(P))
    ll_ora = po.lib().oracle_pass_loglik(po.REF_SSE, n, fp(res), fp(P))
    assert np.float32(ll_ora).tobytes() == np.float32(ll_ref).tobytes()


@pytest.mark.parametrize("n", [1, 2, 777, 4096])
def test_normal_equation_accumulation_is_the_references(n):
    rng = np.random.default_rng(100 + n)
    J = np.ascontiguousarray(rng.normal(size=(n, 12)) * rng.uniform(0.1, 50.0, size=(1, 12)), np.float32)
    alpha = f32(3.5, -0.25, -0.25, 120.0)
    A_ref = np.zeros(36, np.float32)
    A_ora = np.zeros(36, np.float64)
    ref.ref_rank_update_2x6(n, fp(J), fp(alpha), fp(A_ref))
    po.lib().oracle_rank_update_2x6(fp(J), n, fp(alpha), po.REF_SSE, A_ora.ctypes.data_as(C.POINTER(C.c_double)))
    assert A_ora.astype(np.float32).tobytes() == A_ref.tobytes()


def test_intrinsics_scale_is_the_references():
    out = np.zeros(4, np.float32)
    K = f32(517.3, 516.5, 318.6, 255.3)
    ref.ref_intrinsics_scale(fp(K), 0.5, fp(out))
    assert out.tobytes() == (K * np.float32(0.5)).tobytes()     # Q17: offsets halved without half-pixel correction


@pytest.mark.parametrize("w,h", [(640, 480), (100, 76), (17, 5)])
def test_raw_depth_conversion_is_the_references(w, h):
    rng = np.random.default_rng(w)
    raw = rng.integers(0, 65536, (h, w)).astype(np.uint16)
    raw[rng.random((h, w)) < 0.2] = 0                                     # holes -> NaN
    want = np.zeros((h, w), np.float32)
    for sse in ((1, 0) if w % 8 == 0 else (0,)):                          # the SSE version needs whole groups of eight pixels
        ref.ref_convert_raw_depth(sse, raw.ctypes.data_as(C.POINTER(C.c_uint16)), w, h, 1.0 / 5000.0, fp(want))
        got = po.convert_raw_depth(raw)
        assert np.isnan(want).sum() == (raw == 0).sum()
        assert got.tobytes() == want.tobytes()
