"""The sweep of a small pyramid level with the whole current level in LDS (dvo_slam_amd/csrc/align_small.hip, option "small_sweep"): the
default schedule's 80 x 60 / 40 x 30 levels -- what the contracted window sweep does not take -- against the gathering sweep they ran
until round 6 and against the oracle.  Same function (dvo_core/src/dense_tracking_impl.cpp:148-281, dense_tracking.cpp:448-476), a few
ulp apart in the blended gradients: one linearisation within the default schedule's bounds, whole matches to the stopping rule's precision."""
import numpy as np
import pytest

import common as cm
import dvo_slam_amd as d
from dvo_slam_amd import datagen
from oracle import pyoracle as po
from test_gpu_coarse import frames_of

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ctx():
    yield d.Context(0)


@pytest.mark.parametrize("w,h,level", [(640, 480, 3), (320, 240, 2), (320, 240, 3), (1280, 960, 4), (160, 120, 1)])
def test_one_linearisation_of_a_small_level_against_the_oracle_and_the_gathering_sweep(ctx, w, h, level):
    pair = cm.synth(77 + w, w, h)
    cam = d.RgbdCameraPyramid(w, h, pair["K"], ctx)
    cam.build(level + 1)
    gref, gcur = cam.create_raw(pair["grey_ref"], pair["depth_ref"]), cam.create_raw(pair["grey_cur"], pair["depth_cur"])
    oref, ocur = po.pyramids_from_pair(pair, level + 1)
    trk = d.DenseTracker(d.Config(FirstLevel=level, LastLevel=level), ctx)
    T34 = po.se3_exp(np.array([0.004, -0.003, 0.002, 0.005, -0.004, 0.003]))[:3]
    o = po.level_iteration(oref, ocur, level, T34, first=True)
    P_prev = o["P"]
    for first in (True, False):
        o = po.level_iteration(oref, ocur, level, T34, P_prev=None if first else P_prev, first=first)
        out = {}
        for small in (1, 0):
            ctx.set_option("small_sweep", small)
            out[small] = trk.level_iteration(gref, gcur, level, T34, P_prev=None if first else P_prev, first=first)
        g = out[1]
        assert abs(g["n"] - o["n"]) <= max(1, int(1e-4 * o["n"])) and o["n"] >= 50
        scale_A, scale_b = np.abs(o["A"]).max(), np.abs(o["b"]).max()
        slack = 20.0 * abs(g["n"] - o["n"]) / o["n"]
        assert np.abs(g["A"] - o["A"]).max() <= (1e-5 + slack) * scale_A
        assert np.abs(g["b"] - o["b"]).max() <= (1e-5 + slack) * scale_b + 1e-9 * scale_A
        assert np.abs(g["P"] - o["P"]).max() <= (1e-5 + slack) * np.abs(o["P"]).max()
        assert abs(g["neg_ll"] - o["neg_ll"]) <= (2e-5 + slack) * abs(o["neg_ll"])
        assert np.array_equal(g["A"], g["A"].T)
        # ... and the gathering sweep of rounds 1-5 gives the same linearisation to the same bounds
        assert abs(out[0]["n"] - g["n"]) <= max(1, int(1e-4 * o["n"]))
        assert np.abs(out[0]["A"] - g["A"]).max() <= 2e-5 * scale_A and np.abs(out[0]["b"] - g["b"]).max() <= 2e-5 * scale_b + 1e-9 * scale_A


@pytest.mark.parametrize("n,copies", [(6, 25), (5, 120), (4, 1)])
def test_whole_matches_with_and_without_the_small_level_sweep(ctx, n, copies):
    """150 / 600 / 4 pairs of 640 x 480 down to level 0 (the launch path; the 4-pair batch with the resident kernel off): the level-3
    sweep from LDS or gathered -- the same transforms to the stopping rule's precision, the oracle's to 2e-6."""
    w, h = 640, 480
    b = datagen.synth_batch(4100 + n, n, w, h)
    refs, curs = frames_of(ctx, b, w, h, 4, n)
    order = [i % n for i in range(n * copies)]
    refs, curs = [refs[i] for i in order], [curs[i] for i in order]
    trk = d.DenseTracker(d.Config(FirstLevel=3, LastLevel=0), ctx)
    ctx.set_option("resident", 0)
    out = {}
    for small in (0, 1, 1):
        ctx.set_option("small_sweep", small)
        out[small] = trk.match_batch_arrays(refs, curs)
        assert np.isfinite(out[small]["T"]).all()
    assert max(cm.twist_matrix_error(out[1]["T"][k], out[0]["T"][k]) for k in range(len(order))) < 2e-6
    assert np.abs(out[1]["n_iterations"].astype(int) - out[0]["n_iterations"].astype(int)).max() <= 2
    for k in range(n):
        pair = {key: b[key][k] for key in ("grey_ref", "depth_ref", "grey_cur", "depth_cur")}
        pair["K"] = b["K"]
        oref, ocur = po.pyramids_from_pair(pair, 4)
        o = po.match(oref, ocur, po.make_config(3, 0, 100, 5e-7, mode=po.MATH))
        assert cm.twist_matrix_error(out[1]["T"][k], o["T"]) < 2e-6
    # copies of a pair: the same bytes wherever they sit in the batch
    for k, i in enumerate(order):
        assert np.array_equal(out[1]["T"][k], out[1]["T"][i])


def test_frames_carry_plane_c_of_their_small_levels_in_the_flavour_their_batch_reads(ctx):
    """A streamed ingest of a large batch writes plane C alone at 80 x 60 (align_small.hip reads it), a small one both flavours (the
    resident kernel gathers taps); whatever a later batch misses is derived, bit-identically."""
    w, h = 640, 480
    ctx.set_option("small_sweep", 1)
    b = datagen.synth_batch(8, 3, w, h)
    refs, curs = frames_of(ctx, b, w, h, 4, 3)
    trk = d.DenseTracker(d.Config(FirstLevel=3, LastLevel=0), ctx)
    small = trk.match_batch_arrays(refs, curs)                    # 3 pairs: the resident kernel
    order = [i % 3 for i in range(200)]
    many = trk.match_batch_arrays([refs[i] for i in order], [curs[i] for i in order])   # the same frames on the launch path: plane C derived
    for k, i in enumerate(order):
        assert cm.twist_matrix_error(many["T"][k], small["T"][i]) < 2e-6
    again = trk.match_batch_arrays(refs, curs)
    assert np.array_equal(again["T"], small["T"])
