"""CPU tier: the device headers (pixel_math.h, solver_logic.h, se3_device.h), compiled for the host by
tests/emul/emul_device.cpp, against the oracle's MATH mode.  Checks the per-pixel arithmetic and the
Gauss-Newton state machine the gfx950 kernels are built from, without a GPU.  (The reduction scaffolding and
the pyramid kernels are only reachable on the GPU: tests/test_gpu_parity.py.)"""
import ctypes as C

import numpy as np
import pytest
import scipy.linalg

import common as cm
import dvo_slam_amd as d
from oracle import pyoracle as po
from test_oracle import hat6


def test_device_se3_and_solve():
    L = cm.emul_lib()
    rng = np.random.default_rng(0)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))   # noqa: E731
    for scale in (0.0, 1e-9, 1e-6, 1e-3, 0.03, 0.7, 2.0):
        for _ in range(5):
            x = rng.uniform(-1, 1, 6) * scale
            T = np.zeros(16)
            L.emul_se3_exp(dp(x), dp(T))
            assert np.allclose(T.reshape(4, 4), scipy.linalg.expm(hat6(x)), atol=1e-13)
            assert np.allclose(T.reshape(4, 4), po.se3_exp(x), atol=1e-13)     # independent oracle implementation
            y = np.zeros(6)
            L.emul_se3_log(dp(T), dp(y))
            assert np.allclose(y, x, atol=1e-11)
    for _ in range(20):
        M = rng.normal(size=(12, 6))
        A = np.ascontiguousarray(M.T @ M * 10.0 ** rng.uniform(-2, 6))
        b = rng.normal(size=6)
        x = np.zeros(6)
        assert L.emul_solve6(dp(A.reshape(-1)), dp(b), dp(x)) == 0
        assert np.allclose(x, np.linalg.solve(A, b), rtol=1e-9, atol=1e-12)
    x = np.zeros(6)
    assert L.emul_solve6(dp(np.zeros(36)), dp(np.ones(6)), dp(x)) == 1 and np.isnan(x).all()


@pytest.fixture
def schedule():
    """selects which form of the per-pixel stages the emulation runs (tests/emul/emul_device.cpp::g_schedule)"""
    L = cm.emul_lib()
    yield L.emul_set_schedule
    L.emul_set_schedule(0)


@pytest.mark.parametrize("level", [2, 1, 0])
@pytest.mark.parametrize("form", [0, 1])
def test_pixel_math_bit_exact_against_oracle(level, form, schedule):
    """form 0: the staged stage functions; 1: the straight-line stages the matrix-core sweep runs (with the
    short weight / Jacobian forms and Gram accumulation of the sqrt(w)-scaled vector)."""
    schedule(form)
    pair = cm.synth(31, 320, 240)
    ref, cur = cm.oracle_pyramids(pair, 3)
    ep = cm.EmulPair(ref, cur, 3)
    T34 = po.se3_exp(np.array([0.004, -0.003, 0.002, 0.005, -0.004, 0.003]))[:3]
    o = po.level_iteration(ref, cur, level, T34, first=True, mode=po.MATH, want_residuals=True)
    e = ep.level_iteration(level, T34, first=True)
    assert e["n"] == o["n"] and e["n_selected"] == o["n_selected"]
    assert np.array_equal(np.isnan(e["residuals"]), np.isnan(o["residuals"]))
    assert np.array_equal(np.nan_to_num(e["residuals"]), np.nan_to_num(o["residuals"]))     # bit-exact residuals
    assert np.allclose(e["P"], o["P"], rtol=1e-5)
    assert abs(e["neg_ll"] - o["neg_ll"]) <= 1e-7 * abs(o["neg_ll"])
    assert np.abs(e["A"] - o["A"]).max() <= 1e-6 * np.abs(o["A"]).max()
    assert np.abs(e["b"] - o["b"]).max() <= 1e-6 * np.abs(o["b"]).max()
    o2 = po.level_iteration(ref, cur, level, T34, P_prev=o["P"], first=False, mode=po.MATH)
    e2 = ep.level_iteration(level, T34, P_prev=o["P"], first=False)
    assert e2["n"] == o2["n"]
    assert np.abs(e2["A"] - o2["A"]).max() <= 1e-6 * np.abs(o2["A"]).max()
    assert np.abs(e2["b"] - o2["b"]).max() <= 1e-6 * np.abs(o2["b"]).max()



@pytest.mark.parametrize("seed,first,last,mu,init,precision", [
    (1234, 3, 0, 0.0, False, 1e-4), (1234, 3, 0, 0.0, False, 5e-7), (2, 3, 1, 0.05, True, 1e-4), (3, 2, 2, 0.0, False, 5e-7)])
def test_state_machine_against_oracle_driver(seed, first, last, mu, init, precision):
    pair = cm.synth(seed, 320, 240)
    ref, cur = cm.oracle_pyramids(pair, first + 1)
    ep = cm.EmulPair(ref, cur, first + 1)
    cfg = d.Config(FirstLevel=first, LastLevel=last, Mu=mu, UseInitialEstimate=init, Precision=precision)
    T0 = po.se3_exp(0.5 * pair["xi_true"]) if init else None
    e = ep.match(cfg, T0)
    o = po.match(ref, cur, cm.oracle_config_from(cfg, po.MATH), T0)
    s = cm.compare_runs(e, o)
    assert s["n_mismatch"] == 0 and s["max_x_err"] < 2e-5
    assert s["max_iter_count_diff"] <= 2
    assert s["T_err"] < (2e-5 if precision > 1e-6 else 1e-6)
    if s["structure_mismatch"] == 0:
        assert np.abs(e["information"] - o["information"]).max() <= 2e-3 * np.abs(o["information"]).max()
        assert abs(e["loglik"] - o["loglik"]) <= 1e-3 * abs(o["loglik"])
    # keyframe-selection statistics in the result record = the reference's host-side derivations from the same result
    k = cm.keyframe_statistics_from(e)
    assert e["entropy"] == pytest.approx(k["entropy"], rel=1e-12) and e["condition_number"] == pytest.approx(k["condition_number"], rel=1e-9)
    assert e["constraint_ratio"] == k["constraint_ratio"] and e["constraint_ratio_accepted"] == k["constraint_ratio_accepted"]
    ko = cm.keyframe_statistics_from(o)          # ... and agree with the oracle run's
    assert e["entropy"] == pytest.approx(ko["entropy"], abs=2e-2) and e["constraint_ratio"] == pytest.approx(ko["constraint_ratio"], abs=2e-3)


@pytest.mark.parametrize("seed,w,h,first,last,mu,init,precision,cap", [
    (1234, 320, 240, 3, 0, 0.0, False, 5e-7, 100), (1234, 320, 240, 3, 0, 0.0, False, 1e-4, 100), (2, 320, 240, 3, 1, 0.05, True, 1e-4, 50),
    (3, 320, 240, 2, 2, 0.0, False, 5e-7, 100), (9, 131, 97, 2, 0, 0.0, False, 5e-7, 100), (5, 160, 120, 2, 0, 0.5, True, 1e-5, 4),
    (6, 160, 120, 2, 0, 0.0, False, 0.0, 3), (7, 96, 72, 1, 0, 0.0, False, 5e-7, 1)])
def test_speculative_control_flow_of_the_resident_kernel_is_the_state_machine(seed, w, h, first, last, mu, init, precision, cap):
    """The resident kernel runs gn_step before the log-likelihood of a pass is known and settles accept / revert one exchange later
    (gn_commit_loglik; a rejection restores the snapshot the pass started from and replays the pass in full form).  On the host,
    with the same sweeps, that control flow must leave the very same bits as the plain loop: result, level records, iteration
    records -- over levels that end on a rejection, on a small increment, on the iteration cap and on too few constraints."""
    pair = cm.synth(seed, w, h)
    ref, cur = cm.oracle_pyramids(pair, first + 1)
    ep = cm.EmulPair(ref, cur, first + 1)
    cfg = d.Config(FirstLevel=first, LastLevel=last, Mu=mu, UseInitialEstimate=init, Precision=precision, MaxIterationsPerLevel=cap)
    T0 = po.se3_exp(0.5 * pair["xi_true"]) if init else None
    plain = ep.match(cfg, T0, raw=True)
    spec = ep.match(cfg, T0, speculative=True, raw=True)
    assert [(L["id"], len(L["iterations"]), L["termination"]) for L in spec["levels"]] == [(L["id"], len(L["iterations"]), L["termination"]) for L in plain["levels"]]
    assert spec["raw"] == plain["raw"]


def test_speculative_control_flow_without_constraints():
    h, w = 60, 80
    I = np.random.default_rng(0).uniform(0, 255, (h, w)).astype(np.float32)
    Z = np.full((h, w), np.nan, np.float32)
    ref, cur = po.Pyramid(I, Z, po.FR1_K / 8, 2), po.Pyramid(I, Z, po.FR1_K / 8, 2)
    ep = cm.EmulPair(ref, cur, 2)
    cfg = d.Config(FirstLevel=1, LastLevel=0)
    plain, spec = ep.match(cfg, raw=True), ep.match(cfg, speculative=True, raw=True)
    assert [L["termination"] for L in spec["levels"]] == [1, 1] and spec["raw"] == plain["raw"]


@pytest.mark.parametrize("group", [2, 5, 9])
def test_exchange_protocol_of_the_resident_kernel_with_host_threads(group):
    """8-byte {value, sequence number} slots, rows double-buffered by the parity of the exchange count, relaxed stores and polling
    loads: threads at randomly disturbed paces never gather a stale or torn row."""
    assert cm.emul_lib().emul_exchange_stress(group, 1500, 24, 12345 + group) == 0


def test_linear_walk_locates_every_pixel():
    """Row and column of a flat pixel index by one float multiply and two corrections (linear_walk.h), for every index of level sizes
    up to the 2^24-pixel limit the callers enforce."""
    for w, h in ((640, 480), (320, 240), (160, 120), (80, 60), (131, 97), (65, 48), (1280, 960), (1919, 1079), (4096, 4095), (3, 5)):
        assert w * h < (1 << 24)
        assert cm.emul_lib().emul_locate_check(w, h) == 0, (w, h)


def test_sym6_eigenvalues_and_degenerate_statistics():
    L = cm.emul_lib()
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))   # noqa: E731
    rng = np.random.default_rng(4)
    for k in range(40):
        M = rng.normal(size=(8, 6)) * 10.0 ** rng.uniform(-3, 3, size=6)      # badly scaled columns: condition numbers up to ~1e12
        A = np.ascontiguousarray(M.T @ M)
        if k % 5 == 0:
            A -= 0.5 * np.trace(A) / 6 * np.eye(6)                             # indefinite
        ev = np.zeros(6)
        L.emul_sym6_eigenvalues(dp(A.reshape(-1)), dp(ev))
        want = np.linalg.eigvalsh(A)
        assert np.allclose(np.sort(ev), want, rtol=1e-9, atol=1e-12 * np.abs(want).max())
    ev = np.zeros(6)
    L.emul_sym6_eigenvalues(dp(np.full(36, np.nan)), dp(ev))
    assert np.isnan(ev).all()


def test_state_machine_edge_cases():
    # no depth at all: one pass, n = 0, criterion overwritten to IncrementTooSmall, identity out, NaN information
    h, w = 60, 80
    I = np.random.default_rng(0).uniform(0, 255, (h, w)).astype(np.float32)
    Z = np.full((h, w), np.nan, np.float32)
    ref = po.Pyramid(I, Z, po.FR1_K / 8, 1)
    cur = po.Pyramid(I, Z, po.FR1_K / 8, 1)
    ep = cm.EmulPair(ref, cur, 1)
    cfg = d.Config(FirstLevel=0, LastLevel=0)
    e = ep.match(cfg)
    o = po.match(ref, cur, cm.oracle_config_from(cfg, po.MATH))
    assert [L["termination"] for L in e["levels"]] == [L["termination"] for L in o["levels"]] == [1]
    assert e["levels"][0]["iterations"][0]["n"] == 0
    assert np.allclose(e["T"], np.eye(4)) and np.isnan(e["information"]).all()
    # iteration cap
    pair = cm.synth(4, 160, 120)
    ref, cur = cm.oracle_pyramids(pair, 3)
    ep = cm.EmulPair(ref, cur, 3)
    cfg = d.Config(FirstLevel=2, LastLevel=1, MaxIterationsPerLevel=3, Precision=0.0)
    e = ep.match(cfg)
    o = po.match(ref, cur, cm.oracle_config_from(cfg, po.MATH))
    cm.compare_runs(e, o)
    assert all(len(L["iterations"]) <= 3 for L in e["levels"])


from hypothesis import HealthCheck, given, settings, strategies as st   # noqa: E402


@settings(max_examples=30, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
@given(seed=st.integers(0, 10_000), w=st.integers(24, 200), h=st.integers(20, 150), level=st.integers(0, 1), form=st.integers(0, 1),
       xi=st.lists(st.floats(-0.06, 0.06), min_size=6, max_size=6))
def test_random_sizes_and_transforms_bit_exact_against_oracle(seed, w, h, level, form, xi, schedule):
    """any image size (odd ones too), any small transform, both forms of the device stages: valid count and residuals equal the
    oracle's MATH mode bit for bit, sums to accumulation-order tolerance"""
    schedule(form)
    pair = cm.synth(seed, w, h)
    ref, cur = cm.oracle_pyramids(pair, level + 1)
    ep = cm.EmulPair(ref, cur, level + 1)
    T34 = po.se3_exp(np.array(xi))[:3]
    o = po.level_iteration(ref, cur, level, T34, first=True, mode=po.MATH, want_residuals=True)
    e = ep.level_iteration(level, T34, first=True)
    assert e["n"] == o["n"] and e["n_selected"] == o["n_selected"]
    assert np.array_equal(np.isnan(e["residuals"]), np.isnan(o["residuals"]))
    assert np.array_equal(np.nan_to_num(e["residuals"]), np.nan_to_num(o["residuals"]))
    if o["n"] >= 6:
        assert np.abs(e["A"] - o["A"]).max() <= 2e-6 * np.abs(o["A"]).max()
        assert np.abs(e["b"] - o["b"]).max() <= 2e-6 * max(np.abs(o["b"]).max(), 1e-3 * np.abs(o["A"]).max())
