"""CPU tier: pins the oracle (oracle/) against analytic identities, scipy / numpy restatements and the committed
golden fixtures.  The reference has no tests of its own; tests/test_oracle_ref.py pins the oracle's REF_SSE
mode bit for bit against the reference's own translation units (oracle/_ref) -- this file covers what that cannot: SE(3) and
the 6x6 solve (external dependencies of the reference), the MATH mode and its distance to REF_SSE."""
import numpy as np
import pytest
import scipy.linalg

import common as cm
from oracle import pyoracle as po


def hat6(x):
    v, w = x[:3], x[3:]
    M = np.zeros((4, 4))
    M[:3, :3] = [[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]
    M[:3, 3] = v
    return M


@pytest.mark.parametrize("scale", [0.0, 1e-12, 1e-7, 1e-3, 0.03, 0.5, 2.5])
def test_se3_exp_log_against_scipy(scale):
    rng = np.random.default_rng(int(scale * 1e6) + 1)
    for _ in range(5):
        x = rng.uniform(-1, 1, 6) * scale
        T = po.se3_exp(x)
        assert np.allclose(T, scipy.linalg.expm(hat6(x)), atol=1e-13)
        if np.linalg.norm(x[3:]) < 3.0:
            assert np.allclose(po.se3_log(T), x, atol=1e-12)
        assert np.allclose(po.se3_exp(po.se3_log(T)), T, atol=1e-12)   # beyond pi the log is the equivalent shorter rotation
        assert np.allclose(T[:3, :3] @ T[:3, :3].T, np.eye(3), atol=1e-14)


def test_solve6_against_numpy():
    rng = np.random.default_rng(3)
    for _ in range(20):
        M = rng.normal(size=(12, 6))
        A = M.T @ M * 10.0 ** rng.uniform(-2, 6)
        b = rng.normal(size=6)
        x, rc = po.solve6(A, b)
        assert rc == 0
        assert np.allclose(x, np.linalg.solve(A, b), rtol=1e-9, atol=1e-12)


def test_reduce_stage_known_answer():
    """Shape of the reference's only test-like file, dvo_core/src/sse_test.cpp: sum_i J_i^T alpha J_i."""
    g = cm.load_golden("reduce_kat.npz")
    A_math = po.rank_update_2x6(g["J"], g["alpha"], po.MATH)
    A_ref = po.rank_update_2x6(g["J"], g["alpha"], po.REF_SSE)
    scale = np.abs(g["A"]).max()
    assert np.abs(A_math - g["A"]).max() / scale < 1e-12
    assert np.abs(A_ref - g["A"]).max() / scale < 2e-5      # float sequential accumulation over 4096 rows
    assert np.array_equal(A_ref, A_ref.T)


def test_depth_and_grey_ingest():
    raw = np.array([0, 1, 5000, 65535, 12345], np.uint16)
    z = po.convert_raw_depth(raw)
    assert np.isnan(z[0])
    assert np.array_equal(z[1:], raw[1:].astype(np.float32) * np.float32(1.0 / 5000.0))
    bgr = np.array([[[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 200, 30]]], np.uint8)
    g = po.bgr_to_grey(bgr)[0]
    expect = np.round(0.114 * bgr[0, :, 0] + 0.587 * bgr[0, :, 1] + 0.299 * bgr[0, :, 2])
    assert np.abs(g - expect).max() <= 1.0
    assert g[0] == 255 and g[1] == 0


def test_pyramid_model():
    pair = cm.synth(11, 160, 120)
    ref, _ = cm.oracle_pyramids(pair, 3)
    I0, K0 = ref.plane(0, 0)
    Z0, _ = ref.plane(0, 1)
    I1, K1 = ref.plane(1, 0)
    Z1, _ = ref.plane(1, 1)
    assert np.array_equal(K1, K0 * np.float32(0.5))                      # intrinsic_matrix.cpp:90-93 (Q17)
    expect = ((I0[0::2, 0::2] + I0[0::2, 1::2]) + I0[1::2, 0::2] + I0[1::2, 1::2]) / np.float32(4.0)
    assert np.array_equal(I1, expect)                                     # rgbd_image.cpp:38-55
    assert np.array_equal(np.nan_to_num(Z1, nan=-1), np.nan_to_num(Z0[0::2, 0::2], nan=-1))   # :127-139, NaN kept (Q18)
    Ix, _ = ref.plane(0, 2)
    Iy, _ = ref.plane(0, 3)
    assert np.array_equal(Ix[:, 1:-1], (I0[:, 2:] - I0[:, :-2]) * np.float32(0.5))
    assert np.array_equal(Ix[:, 0], (I0[:, 1] - I0[:, 0]) * np.float32(0.5))   # clamped border
    assert np.array_equal(Iy[-1], (I0[-1] - I0[-2]) * np.float32(0.5))
    # selection predicate (point_selection.h:63-66)
    n, mask = ref.select(0)
    Zx, _ = ref.plane(0, 4)
    Zy, _ = ref.plane(0, 5)
    ok = np.isfinite(Z0) & np.isfinite(Zx) & np.isfinite(Zy) & ((np.abs(Ix) > 0) | (np.abs(Iy) > 0) | (np.abs(Zx) > 0) | (np.abs(Zy) > 0))
    assert n == ok.sum() and np.array_equal(mask.astype(bool), ok)
    n_thr, _ = ref.select(0, 5.0, 0.02)
    assert 0 < n_thr < n


def numpy_linearisation(ref, cur, level, T34, P_prev, first):
    """Independent float64 numpy restatement of one MATH-mode linearisation."""
    rp = [ref.plane(level, k)[0].astype(np.float64) for k in range(6)]
    cp = [cur.plane(level, k)[0] for k in range(6)]
    K = ref.plane(level, 0)[1].astype(np.float64)
    fx, fy, ox, oy = K
    h, w = rp[0].shape
    _, mask = ref.select(level)
    v, u = np.mgrid[0:h, 0:w]
    Z = rp[1]
    X = (u - ox) / fx * Z
    Y = (v - oy) / fy * Z
    T = np.asarray(T34, np.float64).reshape(3, 4)
    p = np.stack([X, Y, Z], -1) @ T[:, :3].T + T[:, 3]
    with np.errstate(all="ignore"):
        uu = (fx * p[..., 0] + ox * p[..., 2]) / p[..., 2]
        vv = (fy * p[..., 1] + oy * p[..., 2]) / p[..., 2]
        ok = mask.astype(bool) & (uu >= 0) & (uu <= w - 2) & (vv >= 0) & (vv <= h - 2)
        u0 = np.floor(np.where(ok, uu, 0)).astype(int)
        v0 = np.floor(np.where(ok, vv, 0)).astype(int)
        a = uu - u0
        b = vv - v0
        c = []
        for k in range(6):
            P = cp[k].astype(np.float64)
            c.append((1 - b) * ((1 - a) * P[v0, u0] + a * P[v0, u0 + 1]) + b * ((1 - a) * P[v0 + 1, u0] + a * P[v0 + 1, u0 + 1]))
        ok &= np.all(np.isfinite(np.stack(c)), axis=0)
        r0 = (c[0] - rp[0]) / 255.0
        r1 = c[1] - p[..., 2]
        ok &= r1 > -20.0 * (0.0012 + 0.0019 * (Z - 0.4) ** 2)
    sel = ok
    r = np.stack([r0[sel], r1[sel]], -1)
    n = r.shape[0]
    wgt = np.ones(n) if first else 7.0 / (5.0 + np.einsum("ni,ij,nj->n", r, np.asarray(P_prev, np.float64).reshape(2, 2), r))
    C = np.einsum("n,ni,nj->ij", wgt, r, r) / (n - 3)
    Pm = np.linalg.inv(C)
    ll = 0.5 * n * np.log(np.linalg.det(Pm)) - 3.5 * np.log1p(0.2 * np.einsum("ni,ij,nj->n", r, Pm, r)).sum()
    x, y, z = X[sel], Y[sel], Z[sel]
    zero, one = np.zeros(n), np.ones(n)
    Jw0 = np.stack([1 / z, zero, -x / z**2, -x * y / z**2, 1 + x**2 / z**2, -y / z], -1)
    Jw1 = np.stack([zero, 1 / z, -y / z**2, -(1 + y**2 / z**2), x * y / z**2, x / z], -1)
    Jz = np.stack([zero, zero, one, y, -x, zero], -1)
    gix = 0.5 * fx * (c[2][sel] + rp[2][sel]) / 255.0
    giy = 0.5 * fy * (c[3][sel] + rp[3][sel]) / 255.0
    J0 = gix[:, None] * Jw0 + giy[:, None] * Jw1
    J1 = (fx * c[4][sel])[:, None] * Jw0 + (fy * c[5][sel])[:, None] * Jw1 - Jz
    J = np.stack([J0, J1], 1)
    W = wgt[:, None, None] * Pm
    A = np.einsum("nki,nkl,nlj->ij", J, W, J)
    bb = -np.einsum("nki,nkl,nl->i", J, W, r)
    return dict(n=n, C=C, P=Pm, neg_ll=-ll, A=A, b=bb)


@pytest.mark.parametrize("level,first", [(2, True), (1, False), (0, False)])
def test_oracle_math_matches_numpy_restatement(level, first):
    pair = cm.synth(21, 320, 240)
    ref, cur = cm.oracle_pyramids(pair, 3)
    T34 = po.se3_exp(np.array([0.003, -0.002, 0.004, 0.004, -0.006, 0.002]))[:3]
    Pp = np.array([[60.0, -10.0], [-10.0, 600.0]])
    o = po.level_iteration(ref, cur, level, T34, P_prev=Pp, first=first, mode=po.MATH)
    e = numpy_linearisation(ref, cur, level, T34, Pp, first)
    assert abs(o["n"] - e["n"]) <= max(3, 2e-4 * e["n"])   # float32 vs float64 validity decisions at pixel borders
    assert np.allclose(o["P"], e["P"], rtol=2e-3)
    assert abs(o["neg_ll"] - e["neg_ll"]) / abs(e["neg_ll"]) < 2e-3
    assert np.abs(o["A"] - e["A"]).max() / np.abs(e["A"]).max() < 2e-3
    assert np.abs(o["b"] - e["b"]).max() / np.abs(e["b"]).max() < 5e-3


def test_ref_sse_quirks_are_present():
    """REF_SSE mode must show the order-dependent quirks of the SSE kernels."""
    pair = cm.synth(5, 160, 120)
    ref, cur = cm.oracle_pyramids(pair, 1)
    T34 = po.se3_exp(np.array([0.002, -0.001, 0.001, 0.003, -0.002, 0.001]))[:3]
    m = po.level_iteration(ref, cur, 0, T34, first=True, mode=po.MATH, want_residuals=True)
    s = po.level_iteration(ref, cur, 0, T34, first=True, mode=po.REF_SSE, want_residuals=True)
    # Q1: the 12-bit reciprocal moves u by up to ~0.04 px, so a few taps land in different (NaN) cells
    assert abs(m["n"] - s["n"]) <= 1.5e-2 * m["n"]
    # Q6: with unit weights the buggy covariance is 2 * sum over even-ranked residuals of r r^T / (n-3)
    r = s["residuals"].reshape(-1, 2)
    r = r[np.isfinite(r[:, 0])].astype(np.float64)
    n = r.shape[0]
    assert n == s["n"]
    even = r[0:n - (n % 2):2]
    C_bug = 2.0 * np.einsum("ni,nj->ij", even, even) / (n - 3)
    if n % 2:
        C_bug += np.outer(r[-1], r[-1]) / (n - 3)
    assert np.allclose([C_bug[0, 0], C_bug[0, 1], C_bug[1, 1]], s["cov"], rtol=2e-4)
    C_true = np.einsum("ni,nj->ij", r, r) / (n - 3)
    assert not np.allclose([C_true[0, 0], C_true[1, 1]], s["cov"][[0, 2]], rtol=1e-4)
    # Q7: the log-likelihood ignores the last n mod 50 residuals
    P = s["P"].astype(np.float64)
    q = np.einsum("ni,ij,nj->n", r, P, r)
    m50 = n - n % 50
    ll = 0.5 * n * np.log(np.linalg.det(P)) - 3.5 * np.log1p(0.2 * q[:m50]).sum()
    assert abs(-ll - s["neg_ll"]) / abs(ll) < 1e-5


def test_quirk_by_quirk_modes_span_math_to_ref_sse():
    """mode = QUIRKS | bits (oracle/dvo_oracle.h): no bit is MATH and every bit is REF_SSE, bit for bit, for whole matches and
    for single linearisations -- so the modes in between attribute the distance between the two (profiles/r03_quirk_table.txt).
    And the attribution itself, per match: the approximate reciprocal in the projection (Q1, dense_tracking_impl.cpp:192) is
    what separates the reference from the exact arithmetic."""
    for seed, (w, h) in ((7, (160, 120)), (3, (320, 240))):
        pair = cm.synth(seed, w, h)
        ref, cur = cm.oracle_pyramids(pair, 3)
        T34 = po.se3_exp(np.array([0.002, -0.001, 0.001, 0.003, -0.002, 0.001]))[:3]
        for a, b in ((po.MATH, po.QUIRKS), (po.REF_SSE, po.QUIRKS | po.Q_ALL)):
            x = po.level_iteration(ref, cur, 0, T34, first=False, P_prev=np.array([900.0, 3.0, 3.0, 400.0], np.float32), mode=a, want_residuals=True)
            y = po.level_iteration(ref, cur, 0, T34, first=False, P_prev=np.array([900.0, 3.0, 3.0, 400.0], np.float32), mode=b, want_residuals=True)
            assert x["n"] == y["n"] and np.array_equal(x["A"], y["A"]) and np.array_equal(x["b"], y["b"]) and x["neg_ll"] == y["neg_ll"]
            assert np.array_equal(x["residuals"], y["residuals"], equal_nan=True)
        for kw in (dict(first_level=2, last_level=0), dict(first_level=2, last_level=1, max_iterations=50, precision=1e-4, mu=0.05)):
            run = {m: po.match(ref, cur, po.make_config(mode=m, **kw)) for m in
                   (po.MATH, po.REF_SSE, po.QUIRKS, po.QUIRKS | po.Q_ALL, po.QUIRKS | po.Q_RCP_PROJECTION,
                    po.QUIRKS | po.Q_ALL & ~po.Q_RCP_PROJECTION)}
            for a, b in ((po.MATH, po.QUIRKS), (po.REF_SSE, po.QUIRKS | po.Q_ALL)):
                assert np.array_equal(run[a]["T"], run[b]["T"]) and np.array_equal(run[a]["information"], run[b]["information"])
                assert [len(L["iterations"]) for L in run[a]["levels"]] == [len(L["iterations"]) for L in run[b]["levels"]]
            d = lambda m: cm.twist_matrix_error(run[m]["T"], run[po.REF_SSE]["T"])
            print(seed, kw, "to the reference: MATH %.2e, MATH + Q1p %.2e, all but Q1p %.2e" % (d(po.MATH), d(po.QUIRKS | po.Q_RCP_PROJECTION), d(po.QUIRKS | po.Q_ALL & ~po.Q_RCP_PROJECTION)))
            assert d(po.QUIRKS | po.Q_RCP_PROJECTION) < 0.35 * d(po.MATH)          # Q1p alone closes most of the distance ...
            assert d(po.QUIRKS | po.Q_ALL & ~po.Q_RCP_PROJECTION) > 0.65 * d(po.MATH)   # ... and all the others together do not


def test_match_recovers_true_motion_both_modes():
    pair = cm.synth(1234)
    ref, cur = cm.oracle_pyramids(pair, 4)
    runs = {}
    for mode in (po.MATH, po.REF_SSE):
        cfg = po.make_config(first_level=3, last_level=0, mode=mode)
        runs[mode] = po.match(ref, cur, cfg)
        xi = po.se3_log(runs[mode]["T"])
        assert np.abs(xi - pair["xi_true"]).max() < 5e-5
        assert [L["id"] for L in runs[mode]["levels"]] == [3, 2, 1, 0]
        assert all(L["termination"] in (0, 1, 2) for L in runs[mode]["levels"])
    # REF_SSE-vs-MATH delta of the recovered transform: reported tolerance for "matches the SSE path"
    assert cm.twist_matrix_error(runs[po.MATH]["T"], runs[po.REF_SSE]["T"]) < 5e-5


def test_driver_quirks():
    pair = cm.synth(9, 160, 120)
    ref, cur = cm.oracle_pyramids(pair, 3)
    # Q16/Q12: always completes; information = A_last * 0.008^2 (Q13)
    cfg = po.make_config(first_level=2, last_level=1, mode=po.MATH)
    r = po.match(ref, cur, cfg)
    last = r["levels"][-1]
    it = last["iterations"][-2] if last["termination"] == 2 else last["iterations"][-1]
    assert np.allclose(r["information"], it["A"] * 0.008 * 0.008)
    # max iterations
    cfg = po.make_config(first_level=2, last_level=2, max_iterations=2, precision=0.0, mode=po.MATH)
    r = po.match(ref, cur, cfg)
    assert len(r["levels"][0]["iterations"]) <= 2 and r["levels"][0]["termination"] in (0, 2)
    # initial estimate is applied as the first increment (Q21)
    T0 = po.se3_exp(pair["xi_true"])
    cfg = po.make_config(first_level=2, last_level=0, use_initial_estimate=True, mode=po.MATH)
    r = po.match(ref, cur, cfg, T_init=np.linalg.inv(T0))    # estimate = T^-1 ... start from the inverse on purpose
    assert np.isfinite(r["T"]).all()
    r2 = po.match(ref, cur, cfg, T_init=np.eye(4))
    assert cm.twist_matrix_error(r2["T"], T0) < 2e-3
    # mu > 0 adds mu*I to the information
    cfg = po.make_config(first_level=2, last_level=1, mu=0.05, use_initial_estimate=True, mode=po.MATH)
    r3 = po.match(ref, cur, cfg, T_init=np.eye(4))
    assert np.isfinite(r3["T"]).all() and r3["levels"][0]["iterations"][0]["prior_ll"] >= 0.0


def test_too_few_constraints():
    h, w = 60, 80
    I = np.random.default_rng(0).uniform(0, 255, (h, w)).astype(np.float32)
    Z = np.full((h, w), np.nan, np.float32)
    ref = po.Pyramid(I, Z, po.FR1_K / 8, 1)
    cur = po.Pyramid(I, Z, po.FR1_K / 8, 1)
    r = po.match(ref, cur, po.make_config(first_level=0, last_level=0, mode=po.MATH))
    L = r["levels"][0]
    assert L["valid_pixels"] == 0 and len(L["iterations"]) == 1 and L["iterations"][0]["n"] == 0
    # the reference's post-loop check overwrites TooFewConstraints with IncrementTooSmall because x = log(I) = 0
    assert L["termination"] == 1
    assert np.allclose(r["T"], np.eye(4)) and np.isnan(r["information"]).all()


def test_golden_fixture_regression():
    g = cm.load_golden("s160_seed7.npz")
    pair = dict(grey_ref=g["grey_ref"], depth_ref=g["depth_ref"], grey_cur=g["grey_cur"], depth_cur=g["depth_cur"], K=g["K"])
    ref, cur = po.pyramids_from_pair(pair, 3)
    for name, mode in (("math", po.MATH), ("ref_sse", po.REF_SSE)):
        cfg = po.make_config(first_level=2, last_level=0, max_iterations=100, precision=5e-7, mode=mode)
        run = po.match(ref, cur, cfg)
        rows = g[name + "_iters"]
        k = 0
        for L in run["levels"]:
            for it in L["iterations"]:
                assert (L["id"], it["id"], it["n"]) == tuple(int(v) for v in rows[k, :3])
                assert np.isclose(it["neg_ll"], rows[k, 3], rtol=1e-9)
                assert np.allclose(it["x"], rows[k, 8:14], rtol=1e-7, atol=1e-12, equal_nan=True)
                k += 1
        assert k == rows.shape[0]
        assert np.allclose(run["T"], g[name + "_T"], atol=1e-12)
    # the synthetic generator still produces the stored inputs
    regen = po.synth_pair(7, 160, 120, g["K"])
    for key in ("grey_ref", "depth_ref", "grey_cur", "depth_cur"):
        assert (regen[key] != g[key]).mean() < 1e-3
