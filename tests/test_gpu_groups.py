"""Concurrent pair groups (option "batch_groups", dvo_slam_amd/csrc/capi_groups.inc::run_batch_grouped): a large batch is aligned as two or three
sub-batches at once, the caller's thread on its context, helper threads on twin contexts of the same device -- the way the reference
spreads independent match() calls over the workers of a tbb::parallel_reduce (dvo_slam/src/keyframe_graph.cpp:576-593).  A pair's record
is what its sub-batch gives it: bit-identical to the ungrouped batch wherever both fall into the same schedule class."""
import ctypes as C

import numpy as np
import pytest

import common as cm
import dvo_slam_amd as d
from dvo_slam_amd import _lib, datagen
from test_gpu_coarse import frames_of, raw_match

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ctx():
    yield d.Context(0)


def test_groups_give_every_pair_the_record_of_the_ungrouped_batch(ctx):
    """1200 pairs (ten distinct ones among copies of themselves): one batch, two groups of 600, three of 400 -- all of them batches of
    512 pairs and more without the level hand-over, i.e. one schedule class: every result, level record and iteration record bit for bit."""
    w, h, n = 320, 240, 10
    b = datagen.synth_batch(123, n, w, h)
    refs, curs = frames_of(ctx, b, w, h, 4, n)
    order = [(7 * i + 3) % n for i in range(1200)]
    refs, curs = [refs[i] for i in order], [curs[i] for i in order]
    cfg = d.Config(FirstLevel=3, LastLevel=0)
    ctx.set_option("batch_groups", 1)
    before = ctx.counter("grouped_batches")
    one = raw_match(ctx, cfg, refs, curs)
    assert ctx.counter("grouped_batches") == before
    for groups in (2, 3):
        ctx.set_option("batch_groups", groups)
        for rep in range(2):
            out = raw_match(ctx, cfg, refs, curs)
            assert out[0] == one[0], "results differ with %d groups" % groups
            assert out[1] == one[1], "level records differ with %d groups" % groups
            assert out[2] == one[2], "iteration records differ with %d groups" % groups
    assert ctx.counter("grouped_batches") == before + 4
    # the default: one group (groups that start together stay in phase: no gain measured, include/dvo_hip.h)
    ctx.set_option("batch_groups", 0)
    assert raw_match(ctx, cfg, refs, curs)[:3] == one[:3]
    assert ctx.counter("grouped_batches") == before + 4


def test_small_batches_and_pinned_modes_stay_in_one_group(ctx):
    w, h, n = 320, 240, 6
    b = datagen.synth_batch(321, n, w, h)
    refs, curs = frames_of(ctx, b, w, h, 4, n)
    trk = d.DenseTracker(d.Config(FirstLevel=3, LastLevel=0), ctx)
    order = [i % n for i in range(300)]
    before = ctx.counter("grouped_batches")
    base = trk.match_batch_arrays([refs[i] for i in order], [curs[i] for i in order])
    assert ctx.counter("grouped_batches") == before                       # (the default: one group)
    ctx.set_option("batch_groups", 2)                                     # asked for by name: two groups of 150 (another schedule class:
    two = trk.match_batch_arrays([refs[i] for i in order], [curs[i] for i in order])   # the records agree to the stopping rule's precision)
    assert ctx.counter("grouped_batches") == before + 1
    assert max(cm.twist_matrix_error(two["T"][k], base["T"][k]) for k in range(len(order))) < 2e-6
    assert np.array_equal(two["n_iterations"][:150], two["n_iterations"][[k for k in range(150)]])
    for key in ("deterministic", "ref_compat"):                           # one schedule per pair / the reciprocal table lives in the owner
        ctx.set_option(key, 1)
        trk.match_batch_arrays([refs[i] for i in order], [curs[i] for i in order])
        assert ctx.counter("grouped_batches") == before + 1
        ctx.set_option(key, 0)


def test_statistics_arrays_too_small_are_reported_from_any_group(ctx):
    """DVO_HIP_ERR_CAPACITY (the results are valid) whichever group's slice was truncated."""
    w, h, n = 320, 240, 4
    b = datagen.synth_batch(55, n, w, h)
    refs, curs = frames_of(ctx, b, w, h, 4, n)
    order = [i % n for i in range(260)]
    refs, curs = [refs[i] for i in order], [curs[i] for i in order]
    cfg = d.Config(FirstLevel=3, LastLevel=0)
    for r, c in zip(refs, curs):
        r.build(4); c.build(4)
    ctx.set_option("batch_groups", 2)
    m = len(order)
    cres = (_lib.Result * m)()
    for i in range(m):
        for k, v in enumerate(np.eye(4).reshape(-1)):
            cres[i].transformation[k] = v
    vp = C.c_void_p
    rp = (vp * m)(*[p.ptr for p in refs]); cp = (vp * m)(*[p.ptr for p in curs])
    levels = (_lib.LevelStats * (m * 4))()
    iters = (_lib.IterationStats * (m * 3))()                              # three iteration records per pair: too few
    ccfg = cfg.to_c()
    rc = ctx._lib.dvo_hip_match_batch(ctx.ptr, m, rp, cp, C.byref(ccfg), cres, levels, 4, iters, 3)
    assert rc == _lib.ERR_CAPACITY
    T = np.array([np.array(cres[i].transformation).reshape(4, 4) for i in range(m)])
    assert np.isfinite(T).all()
    for k, i in enumerate(order):
        assert np.array_equal(T[k], T[i])
