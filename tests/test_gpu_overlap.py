"""The slow lane of a batch (capi_schedule.inc::run_batch, option "overlap_tails"): once a few pairs of a large batch are left on a level, they leave the
batch's launch chain for good -- the chain goes on to the next level without them, and a second stream runs them to the end of the match with
launches over a list of pairs (LevelGeom::pair_list), level after level behind the chain, taking up the stragglers of the later levels on the
way.  Every pair runs its levels on its own, as the reference's match() calls do (dvo_core/src/dense_tracking.cpp:200-357).

A pair's arithmetic does not know in which launch it runs, so the bar is BIT identity with the synchronous chain: every byte of every
result, level record and iteration record."""
import ctypes as C

import numpy as np
import pytest

import common as cm
import dvo_slam_amd as d
from dvo_slam_amd import datagen
from oracle import pyoracle as po

from test_gpu_coarse import frames_of, raw_match

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ctx():
    c = d.Context(0)
    yield c


def spread_batch(seed, n_distinct, n, w, h):
    """n pairs made of n_distinct synthetic ones whose motions differ in size (so that they need different numbers of passes), dealt so
    that neighbours in the batch differ"""
    b = datagen.synth_batch(seed, n_distinct, w, h)
    order = [(7 * i) % n_distinct for i in range(n)]
    return b, order


@pytest.mark.parametrize("w,h,first,last,n,fraction,extra", [
    (320, 240, 3, 0, 300, 8, {}),     # beyond the solver steps' hand-over (256 pairs): levels 3, 2, 1 may shed their stragglers
    (320, 240, 3, 1, 272, 4, {}),     # the front end's levels: the last one (1) sheds nothing
    (640, 480, 3, 0, 264, 2, {}),     # BASELINE config 4's shape, shed early (half of the pairs still on the level); the lane's level 0 has
                                      # its log-likelihood pass in a launch of its own
    (320, 240, 3, 0, 300, 2, dict(MaxIterationsPerLevel=8)),                      # levels that end on the iteration cap, in the chain and in the lane
    (320, 240, 3, 1, 288, 2, dict(Mu=0.05, Precision=1e-4, MaxIterationsPerLevel=50, UseInitialEstimate=True)),   # benchmark.yaml: motion prior, initial estimate
])
def test_the_slow_lane_leaves_the_synchronous_chain_s_records(ctx, w, h, first, last, n, fraction, extra):
    b, order = spread_batch(40 + w, 24, n, w, h)
    refs0, curs0 = frames_of(ctx, b, w, h, first + 1, 24)
    refs, curs = [refs0[i] for i in order], [curs0[i] for i in order]
    cfg = d.Config(FirstLevel=first, LastLevel=last, **extra)
    # (initial estimates of very different quality, so that the pairs need different numbers of passes under the loose stopping rule too)
    T0 = [po.se3_exp((0.95 if k % 3 == 0 else 0.0) * np.asarray(b["xi_true"][i])) for k, i in enumerate(order)] if extra.get("UseInitialEstimate") else None
    ctx.set_option("overlap_tails", 0)
    base = raw_match(ctx, cfg, refs, curs, T0)
    before = ctx.counter("overlapped_tails")
    ctx.set_option("overlap_tails", 1)
    ctx.set_option("overlap_fraction", fraction)
    over = raw_match(ctx, cfg, refs, curs, T0)
    assert ctx.counter("overlapped_tails") > before, "no level shed a pair: the test does not test"
    assert ctx.counter("overlapped_steps") > 0
    assert over[0] == base[0], "results differ"
    assert over[1] == base[1], "level records differ"
    assert over[2] == base[2], "iteration records differ"
    # again: the buffers and status words of a tail are reused from batch to batch
    again = raw_match(ctx, cfg, refs, curs, T0)
    assert again[:3] == base[:3]
    # ... and they are real alignments
    if last == 0 and not extra:
        pair = {k: b[k][order[0]] for k in ("grey_ref", "depth_ref", "grey_cur", "depth_cur")}
        pair["K"] = b["K"]
        oref, ocur = po.pyramids_from_pair(pair, first + 1)
        o = po.match(oref, ocur, po.make_config(first, last, 100, 5e-7, mode=po.MATH))
        T = np.array(over[3][0].transformation).reshape(4, 4)
        assert cm.twist_matrix_error(T, o["T"]) < 2e-6


def test_batches_the_overlap_does_not_take(ctx):
    """Off by default... and switched on it leaves alone: batches whose levels are handed over by the solver steps (up to 256 pairs), the
    "deterministic" schedule (the exact window sweep has no pair list), a single level."""
    w, h = 320, 240
    b, order = spread_batch(9, 12, 300, w, h)
    refs0, curs0 = frames_of(ctx, b, w, h, 4, 12)
    refs, curs = [refs0[i] for i in order], [curs0[i] for i in order]
    trk = d.DenseTracker(d.Config(FirstLevel=3, LastLevel=0), ctx)
    before = ctx.counter("overlapped_tails")
    base = trk.match_batch_arrays(refs, curs)
    assert ctx.counter("overlapped_tails") == before
    ctx.set_option("overlap_tails", 1)
    small = trk.match_batch_arrays(refs[:200], curs[:200])
    assert ctx.counter("overlapped_tails") == before
    for k in range(200):
        assert np.array_equal(small["T"][k], base["T"][k]) or cm.twist_matrix_error(small["T"][k], base["T"][k]) < 2e-6
    one = d.DenseTracker(d.Config(FirstLevel=2, LastLevel=2), ctx).match_batch_arrays(refs, curs)
    assert ctx.counter("overlapped_tails") == before and not np.isnan(one["T"]).any()
    ctx.set_option("deterministic", 1)
    det = trk.match_batch_arrays(refs, curs)
    assert ctx.counter("overlapped_tails") == before
    assert max(cm.twist_matrix_error(det["T"][k], base["T"][k]) for k in range(300)) < 2e-6
    ctx.set_option("deterministic", 0)
    on = trk.match_batch_arrays(refs, curs)
    assert ctx.counter("overlapped_tails") > before
    for k in range(300):
        assert np.array_equal(on["T"][k], base["T"][k]) and on["n_iterations"][k] == base["n_iterations"][k]


def test_active_pair_lists_leave_the_records_alone(ctx):
    """Option "tail_lists": the last steps of a level launched over the list of the pairs still on it.  "tail_speculation" 2 makes every
    level one on which the host waits for a step's outcome once few pairs are left (by default only sweeps of 131 072 workgroups and more:
    the finest level of 437 and more 640 x 480 pairs), so that a batch of a test's size gets its lists."""
    w, h, n = 320, 240, 300
    b, order = spread_batch(5, 75, n, w, h)                             # (many distinct pairs: the levels thin out pair by pair)
    refs0, curs0 = frames_of(ctx, b, w, h, 4, 75)
    refs, curs = [refs0[i] for i in order], [curs0[i] for i in order]
    cfg = d.Config(FirstLevel=3, LastLevel=0)
    ctx.set_option("tail_speculation", 2)
    base = raw_match(ctx, cfg, refs, curs)
    before = ctx.counter("listed_steps")
    ctx.set_option("tail_lists", 1)
    listed = raw_match(ctx, cfg, refs, curs)
    assert ctx.counter("listed_steps") > before
    assert listed[:3] == base[:3]
    # with the slow lane beside it: the chain's list leaves the lane's pairs out
    ctx.set_option("overlap_tails", 1)
    ctx.set_option("overlap_fraction", 4)
    both = raw_match(ctx, cfg, refs, curs)
    assert both[:3] == base[:3]
