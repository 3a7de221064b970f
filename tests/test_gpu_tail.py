"""The Gauss-Newton step in the sweep's launch (dvo_slam_amd/csrc/solver_step.h, option "sweep_tail"): on the levels whose log-likelihood
pass runs inside the solver step, the workgroup that completes the LAST tile of a pair runs the pair's step right there -- one launch per
iteration instead of two (the reference's loop body follows its residual pass without leaving the thread either,
dvo_core/src/dense_tracking.cpp:240-357).  The step is the stand-alone kernel's own code (solver_step_body), the tiles' rows and residual
pairs reach it through write-through stores and an arrival counter per pair: every byte of every result, level record and iteration
record must be what the two-launch form gives -- under load, with ragged batches, with pairs that leave their levels at different times."""
import ctypes as C

import numpy as np
import pytest

import common as cm
import dvo_slam_amd as d
from dvo_slam_amd import _lib, datagen
from oracle import pyoracle as po
from test_gpu_coarse import frames_of, raw_match

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ctx():
    yield d.Context(0)


@pytest.mark.parametrize("w,h,first,last,mu,init,precision,n,copies", [
    (640, 480, 3, 0, 0.0, False, 5e-7, 5, 1),       # BASELINE config 4's shape: levels 3 (gathered taps) and 2 (window sweep) end with the step
    (640, 480, 3, 2, 0.0, False, 5e-7, 3, 1),       # the match ends on a level with a tail: results written behind it
    (640, 480, 3, 1, 0.05, True, 1e-4, 4, 1),       # benchmark.yaml: motion prior, initial estimate
    (320, 240, 3, 0, 0.0, False, 5e-7, 6, 30),      # 180 pairs: the launch path with the level hand-over in the steps (<= 256 pairs)
    (320, 240, 3, 0, 0.0, False, 5e-7, 7, 80),      # 560 pairs: no hand-over, two-wavefront solver steps, levels up to 320 x 240 fused
    (262, 194, 2, 0, 0.0, False, 5e-7, 3, 1),       # ragged: 65 x 48 and 131 x 97 (odd widths: gathered, pixels in linear order)
    (1280, 960, 4, 0, 0.0, False, 1e-4, 2, 1),      # BASELINE config 5
])
def test_step_in_the_sweep_s_launch_gives_the_two_launch_records_bit_for_bit(ctx, w, h, first, last, mu, init, precision, n, copies):
    b = datagen.synth_batch(500 + w, n, w, h)
    refs, curs = frames_of(ctx, b, w, h, first + 1, n)
    order = [i % n for i in range(n * copies)]
    refs, curs = [refs[i] for i in order], [curs[i] for i in order]
    cfg = d.Config(FirstLevel=first, LastLevel=last, Mu=mu, UseInitialEstimate=init, Precision=precision, MaxIterationsPerLevel=50 if init else 100)
    T0 = [po.se3_exp(0.5 * np.asarray(b["xi_true"][i])) for i in order] if init else None
    ctx.set_option("resident", 0)
    ctx.set_option("small_sweep", 0)                   # (the gathering sweep on the small levels: the one with a tail)
    ctx.set_option("sweep_tail", 0)
    before = ctx.counter("tail_steps")
    base = raw_match(ctx, cfg, refs, curs, T0)
    assert ctx.counter("tail_steps") == before
    for mode in (1, 2):                                     # 1: the whole step in the tail; 2: its wide half, the serial half in a launch behind it
        ctx.set_option("sweep_tail", mode)
        for rep in range(3 if copies > 1 else 1):          # (a hand-off that is stale one time in a hundred shows up in a few hundred pairs x iterations)
            tail = raw_match(ctx, cfg, refs, curs, T0)
            assert ctx.counter("tail_steps") > before
            assert tail[0] == base[0], "results differ (mode %d)" % mode
            assert tail[1] == base[1], "level records differ (mode %d)" % mode
            assert tail[2] == base[2], "iteration records differ (mode %d)" % mode


def test_pairs_that_leave_their_levels_at_different_times(ctx):
    """An identical pair (leaves every level after its first passes), ordinary pairs, a pair without a single selected pixel, among copies
    of themselves: the arrival counters of the pairs that are no longer on the level still run, their hand-over happens once."""
    w, h = 640, 480
    b = datagen.synth_batch(900, 3, w, h)
    cam = d.RgbdCameraPyramid(w, h, b["K"], ctx)
    cam.build(4)
    flat = np.full((h, w), 128, np.uint8)                                # no gradient: nothing is selected
    fr = [cam.create_raw(b["grey_ref"][0], b["depth_ref"][0]), cam.create_raw(b["grey_ref"][1], b["depth_ref"][1]),
          cam.create_raw(flat, b["depth_ref"][2]), cam.create_raw(b["grey_ref"][2], b["depth_ref"][2])]
    fc = [cam.create_raw(b["grey_ref"][0], b["depth_ref"][0]), cam.create_raw(b["grey_cur"][1], b["depth_cur"][1]),
          cam.create_raw(flat, b["depth_cur"][2]), cam.create_raw(b["grey_cur"][2], b["depth_cur"][2])]
    order = [(3 * i + 1) % 4 for i in range(150)]
    refs, curs = [fr[i] for i in order], [fc[i] for i in order]
    cfg = d.Config(FirstLevel=3, LastLevel=0)
    ctx.set_option("sweep_tail", 0)
    base = raw_match(ctx, cfg, refs, curs)
    for mode in (1, 2):
        ctx.set_option("sweep_tail", mode)
        for rep in range(3):
            assert raw_match(ctx, cfg, refs, curs)[:3] == base[:3]
    # a pair's record does not depend on the batch's other pairs' tails: alone (launch path forced) it is the same bytes
    ctx.set_option("resident", 0)
    size = C.sizeof(_lib.Result)
    ctx.set_option("rows_per_wave", 1)                                   # (the tile height the 150-pair batch got on its gathering level)
    for k in range(4):
        alone = raw_match(ctx, cfg, refs[k:k + 1], curs[k:k + 1])
        assert cm.twist_matrix_error(np.array(alone[3][0].transformation).reshape(4, 4), np.array(base[3][k].transformation).reshape(4, 4)) < 2e-6


def test_the_tail_is_opt_in_and_modes_without_an_instantiation_do_not_use_it(ctx):
    w, h, n = 320, 240, 5
    b = datagen.synth_batch(77, n, w, h)
    refs, curs = frames_of(ctx, b, w, h, 4, n)
    trk = d.DenseTracker(d.Config(FirstLevel=3, LastLevel=0), ctx)
    order = [i % n for i in range(160)]
    before = ctx.counter("tail_steps")
    base = trk.match_batch_arrays([refs[i] for i in order], [curs[i] for i in order])
    assert ctx.counter("tail_steps") == before                          # (off by default: measured slower)
    ctx.set_option("sweep_tail", 1)
    out = trk.match_batch_arrays([refs[i] for i in order], [curs[i] for i in order])
    used = ctx.counter("tail_steps")
    assert used > before
    assert all(np.array_equal(out[key], base[key], equal_nan=True) for key in ("T", "information", "loglik", "n_iterations"))
    for key in ("deterministic", "ref_compat"):
        ctx.set_option(key, 1)
        out = trk.match_batch_arrays([refs[i] for i in order], [curs[i] for i in order])
        assert ctx.counter("tail_steps") == used
        assert max(cm.twist_matrix_error(out["T"][k], base["T"][k]) for k in range(len(order))) < (1e-4 if key == "ref_compat" else 2e-6)
        ctx.set_option(key, 0)
