"""The resident match kernel (dvo_slam_amd/csrc/align_resident.hip: a whole match, or its coarse levels, in one launch; every pair
owned by a group of workgroups) against the launch-per-step path of the same library and against the oracle.

The two paths run the same per-pixel arithmetic and the same state machine (solver_logic.h) but add the per-pixel contributions in
a different order, so they agree like two batch-size classes of the launch path do: constraint counts exactly, increments and
results to the precision of the stopping rule; an iteration more or less at the noise floor is possible, not seen on these inputs."""
import os

import numpy as np
import pytest

import common as cm
import dvo_slam_amd as d
from dvo_slam_amd import datagen
from oracle import pyoracle as po
from test_gpu_parity import gpu_pyramids, run_gpu_match

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ctx(gpu_ctx):
    # the resident kernel accumulates its normal equations in f32: the launch path it is compared with runs the f32 schedule too
    # (variant 6; the default's f16 Gram differs in the 7th digit, which moves a termination test on the edge now and then)
    gpu_ctx.set_option("variant", 6)
    yield gpu_ctx
    for key, value in (("variant", 8), ("resident", -1), ("resident_group", 0), ("resident_rows", 0), ("resident_flags", 0), ("resident_cooperative", 0)):
        gpu_ctx.set_option(key, value)


def stats_of(ctx, cfg, refs, curs, T0=None):
    trk = d.DenseTracker(cfg, ctx)
    results = [d.Result() for _ in refs]
    if T0 is not None:
        for r in results:
            r.Transformation = T0.copy()
    trk.match_batch(refs, curs, results, with_stats=True)
    return results


def same_structure(a, b):
    return ([L.Id for L in a.Statistics.Levels] == [L.Id for L in b.Statistics.Levels]
            and [len(L.Iterations) for L in a.Statistics.Levels] == [len(L.Iterations) for L in b.Statistics.Levels]
            and [L.TerminationCriterion for L in a.Statistics.Levels] == [L.TerminationCriterion for L in b.Statistics.Levels])


@pytest.mark.parametrize("seed,w,h,first,last,mu,init,precision", [
    (1234, 640, 480, 3, 0, 0.0, False, 5e-7),     # BASELINE config 2
    (5, 640, 480, 3, 1, 0.05, True, 1e-4),        # benchmark.yaml: prior, initial estimate
    (9, 131, 97, 2, 0, 0.0, False, 5e-7),         # ragged sizes
    (4321, 1280, 960, 4, 0, 0.0, False, 1e-4),    # BASELINE config 5 (level 0: 19200 segments, residuals spill to scratch)
])
@pytest.mark.parametrize("group", [0, 1, 2, 16])
def test_resident_match_equals_launch_path_and_oracle(ctx, seed, w, h, first, last, mu, init, precision, group):
    pair = cm.synth(seed, w, h)
    gref, gcur = gpu_pyramids(ctx, pair, first + 1)
    cfg = d.Config(FirstLevel=first, LastLevel=last, Mu=mu, UseInitialEstimate=init, Precision=precision, MaxIterationsPerLevel=50 if init else 100)
    T0 = po.se3_exp(0.5 * pair["xi_true"]) if init else None
    ctx.set_option("resident", 0)
    base = stats_of(ctx, cfg, [gref], [gcur], T0)[0]
    before = ctx.counter("resident_launches")
    ctx.set_option("resident", 1)                                          # every level, whatever the sweep length
    ctx.set_option("resident_group", group)
    res = stats_of(ctx, cfg, [gref], [gcur], T0)[0]
    assert ctx.counter("resident_launches") == before + 1 and ctx.counter("resident_timeouts") == 0
    shape = lambda r: [(L.Id, len(L.Iterations), L.TerminationCriterion) for L in r.Statistics.Levels]
    # the same passes; the verdict on the LAST pass of a level may differ where successive log-likelihoods are closer than the two
    # paths' rounding (the 131x97 case: pass 9 of level 1 is "accepted, increment too small" on one path, "rejected" on the other)
    assert [x[:2] for x in shape(res)] == [x[:2] for x in shape(base)], (shape(res), shape(base))
    if seed != 9:
        assert same_structure(res, base), (shape(res), shape(base))
    worst_n = worst_ll = worst_x = 0.0
    for La, Lb in zip(res.Statistics.Levels, base.Statistics.Levels):
        assert La.ValidPixels == Lb.ValidPixels and La.MaxValidPixels == Lb.MaxValidPixels
        if La.Id == first:
            assert La.Iterations[0].ValidConstraints == Lb.Iterations[0].ValidConstraints   # same estimate, same arithmetic: exact
        for ia, ib in zip(La.Iterations, Lb.Iterations):
            worst_n = max(worst_n, abs(ia.ValidConstraints - ib.ValidConstraints))
            if np.isfinite(ib.TDistributionLogLikelihood):
                worst_ll = max(worst_ll, abs(ia.TDistributionLogLikelihood - ib.TDistributionLogLikelihood) / abs(ib.TDistributionLogLikelihood))
            if np.isfinite(ib.EstimateIncrement).all() and np.isfinite(ia.EstimateIncrement).all():
                worst_x = max(worst_x, np.abs(ia.EstimateIncrement - ib.EstimateIncrement).max())
    dT = cm.twist_matrix_error(res.Transformation, base.Transformation)
    print("group %d: constraint counts differ by at most %d, -ll by %.1e relative, increments by %.1e, result by %.1e" % (group, worst_n, worst_ll, worst_x, dT))
    # the estimates of the two paths drift apart by rounding (1e-7 per pass); the log-likelihood moves with the estimate
    assert worst_n <= 4 and worst_ll < 1e-4 and worst_x < 5e-6
    assert dT < 2e-6
    assert np.abs(res.Information - base.Information).max() <= 1e-3 * np.abs(base.Information).max()
    assert res.LogLikelihood == pytest.approx(base.LogLikelihood, rel=1e-5)
    # and against the oracle, with the bounds of test_full_match_against_oracle
    oref, ocur = cm.oracle_pyramids(pair, first + 1)
    o = po.match(oref, ocur, cm.oracle_config_from(cfg, po.MATH), T0)
    assert cm.twist_matrix_error(res.Transformation, o["T"]) < (2e-5 if precision > 1e-6 else 1e-6)


@pytest.mark.parametrize("seed,w,h,first,last,mu,init,precision", [
    (1234, 640, 480, 3, 0, 0.0, False, 5e-7),     # BASELINE config 2
    (5, 640, 480, 3, 1, 0.05, True, 1e-4),        # the front end's configuration (benchmark.yaml)
])
@pytest.mark.parametrize("group", [0, 1, 4])
def test_reference_compatible_arithmetic_on_the_resident_kernel(seed, w, h, first, last, mu, init, precision, group):
    """Option ref_compat (projection and weights multiply with the host CPU's _mm_rcp_ps, like the reference's SSE path) on the latency
    path: since round 4 the resident kernel carries that arithmetic too (rounds 2-3 sent every ref_compat match to the launch path), so
    the reference's front-end pattern -- one or two pairs per frame (dvo_slam/src/local_tracker.cpp:157-216) -- has a
    reference-compatible one-launch match.  Against the launch path in the same mode: the same passes, the same constraint counts on
    a level's first pass, results to the precision of the stopping rule; against the reference's own match(): the distance of the
    launch path's ref_compat results (a fraction of the default mode's, tests/test_gpu_replay.py)."""
    ctx = d.Context(0)
    ctx.set_option("ref_compat", 1)
    ctx.set_option("variant", 6)                   # (the f32 Gram on the launch path, like the resident kernel's: see the fixture above)
    pair = cm.synth(seed, w, h)
    gref, gcur = gpu_pyramids(ctx, pair, first + 1)
    cfg = d.Config(FirstLevel=first, LastLevel=last, Mu=mu, UseInitialEstimate=init, Precision=precision, MaxIterationsPerLevel=50 if init else 100)
    T0 = po.se3_exp(0.5 * pair["xi_true"]) if init else None
    ctx.set_option("resident", 0)
    base = stats_of(ctx, cfg, [gref], [gcur], T0)[0]
    before = ctx.counter("resident_launches")
    ctx.set_option("resident", 1)
    ctx.set_option("resident_group", group)
    res = stats_of(ctx, cfg, [gref], [gcur], T0)[0]
    assert ctx.counter("resident_launches") == before + 1 and ctx.counter("resident_timeouts") == 0
    shape = lambda r: [(L.Id, len(L.Iterations)) for L in r.Statistics.Levels]
    assert shape(res) == shape(base), (shape(res), shape(base))
    for La, Lb in zip(res.Statistics.Levels, base.Statistics.Levels):
        if La.Id == first:
            assert La.Iterations[0].ValidConstraints == Lb.Iterations[0].ValidConstraints   # same estimate, same arithmetic: exact
    dT = cm.twist_matrix_error(res.Transformation, base.Transformation)
    oref, ocur = cm.oracle_pyramids(pair, first + 1)
    r = po.match(oref, ocur, cm.oracle_config_from(cfg, po.REF_SSE), T0)            # = the reference's own match(), bit for bit
    d_ref, d_ref_base = cm.twist_matrix_error(res.Transformation, r["T"]), cm.twist_matrix_error(base.Transformation, r["T"])
    ctx.set_option("ref_compat", 0)
    ctx.set_option("resident", 0)
    exact = stats_of(ctx, cfg, [gref], [gcur], T0)[0]
    d_exact = cm.twist_matrix_error(exact.Transformation, r["T"])
    print("group %d: resident vs launch path (both ref_compat) %.1e; to the reference's match(): resident %.1e, launch path %.1e, exact arithmetic %.1e"
          % (group, dT, d_ref, d_ref_base, d_exact))
    assert dT < 2e-6
    assert d_ref < d_ref_base + 2e-6
    if precision < 1e-6:
        assert d_ref < 0.5 * d_exact + 1e-6       # (at Precision 1e-4 two runs of any two arithmetics stop up to 1e-4 apart)


def test_default_policy_uses_the_resident_kernel_for_small_batches_only(ctx):
    b = datagen.synth_batch(3, 40, 320, 240)
    cam = d.RgbdCameraPyramid(320, 240, b["K"], ctx)
    cam.build(3)
    refs = [cam.create_raw(b["grey_ref"][i], b["depth_ref"][i]) for i in range(40)]
    curs = [cam.create_raw(b["grey_cur"][i], b["depth_cur"][i]) for i in range(40)]
    cfg = d.Config(FirstLevel=2, LastLevel=0, Precision=5e-7)
    ctx.set_option("resident", 0)
    base = stats_of(ctx, cfg, refs, curs)
    ctx.set_option("resident", -1)
    for n in (1, 2, 5, 40):                                               # 40 pairs: groups of 4, the finest level stays on the launch path
        before = ctx.counter("resident_launches")
        res = stats_of(ctx, cfg, refs[:n], curs[:n])
        assert ctx.counter("resident_launches") == before + 1
        for a, bb in zip(res, base):
            assert same_structure(a, bb) or abs(sum(len(L.Iterations) for L in a.Statistics.Levels) - sum(len(L.Iterations) for L in bb.Statistics.Levels)) <= 2
            assert cm.twist_matrix_error(a.Transformation, bb.Transformation) < 5e-6
    # a pair's result does not depend on its neighbours in the batch, given the group size (2 and 3 pairs: 64 workgroups each)
    two, five = stats_of(ctx, cfg, refs[:2], curs[:2]), stats_of(ctx, cfg, refs[:3], curs[:3])
    for a, bb in zip(two, five):
        assert np.array_equal(a.Transformation, bb.Transformation) and np.array_equal(a.Information, bb.Information)


def test_default_policy_runs_the_first_level_of_a_mid_size_batch_resident(ctx):
    """Between 1/4 and 7/16 as many pairs as compute units (65 ... 112 on an MI355X) the first level alone runs in the resident
    kernel, two workgroups per pair (they leave an eighth of the chip free), and the launch path takes over from the second level;
    above that no resident launch at all (plan_resident, capi.hip).  Same passes and results as the launch path."""
    n_all = 136
    b = datagen.synth_batch(5, n_all, 160, 120)
    cam = d.RgbdCameraPyramid(160, 120, b["K"], ctx)
    cam.build(3)
    refs = [cam.create_raw(b["grey_ref"][i], b["depth_ref"][i]) for i in range(n_all)]
    curs = [cam.create_raw(b["grey_cur"][i], b["depth_cur"][i]) for i in range(n_all)]
    cfg = d.Config(FirstLevel=2, LastLevel=0, Precision=5e-7)
    ctx.set_option("resident", 0)
    base = stats_of(ctx, cfg, refs, curs)
    ctx.set_option("resident", -1)
    for n, launches in ((72, 1), (112, 1), (128, 0), (n_all, 0)):
        before = ctx.counter("resident_launches")
        res = stats_of(ctx, cfg, refs[:n], curs[:n])
        assert ctx.counter("resident_launches") == before + launches and ctx.counter("resident_timeouts") == 0, n
        for a, bb in zip(res, base):                 # (a pass more or less where an increment sits on the stopping rule's threshold)
            assert same_structure(a, bb) or abs(sum(len(L.Iterations) for L in a.Statistics.Levels) - sum(len(L.Iterations) for L in bb.Statistics.Levels)) <= 4
            assert cm.twist_matrix_error(a.Transformation, bb.Transformation) < 5e-6


def test_frames_streamed_without_their_taps_get_the_first_level_resident_only(gpu_ctx):
    """Round 5: a role-aware ingest of an eighth as many frames as compute units or more (32 on an MI355X) writes plane C alone on the
    levels the window sweep reads (eager_current_flavor) -- and the match of such frames runs the FIRST level in the resident kernel
    and the others on the launch path instead of having the taps derived (plan_resident looks at what the frames hold).  Frames that
    were created without a role get the coarse levels resident as before.  Same passes, same results either way.  (The default
    schedule, variant 8: under the fixture's variant 6 a 160-pixel level is not the window sweep's.)"""
    ctx = gpu_ctx
    n, w, h = 40, 640, 480
    b = datagen.synth_batch(9, n, w, h, nthreads=8)
    cam = d.RgbdCameraPyramid(w, h, b["K"], ctx)
    cam.build(4)
    cfg = d.Config(FirstLevel=3, LastLevel=1, Precision=5e-7)
    fresh = lambda: ([cam.create_raw(b["grey_ref"][i], b["depth_ref"][i]) for i in range(n)],
                     [cam.create_raw(b["grey_cur"][i], b["depth_cur"][i]) for i in range(n)])
    ctx.set_option("resident", 0)
    base = stats_of(ctx, cfg, *fresh())
    ctx.set_option("resident", -1)
    lazy_refs, lazy_curs = fresh()                                    # (no role planes yet: the match has built what its plan reads)
    k0 = ctx.counter("resident_launches"), ctx.counter("resident_levels")
    lazy = stats_of(ctx, cfg, lazy_refs, lazy_curs)
    assert (ctx.counter("resident_launches"), ctx.counter("resident_levels")) == (k0[0] + 1, k0[1] + 2)     # levels 3 and 2
    streamed_refs, streamed_curs = fresh()
    d.update_raw_host_batch(streamed_refs, list(b["grey_ref"]), list(b["depth_ref"]), role="reference", config=cfg)
    d.update_raw_host_batch(streamed_curs, list(b["grey_cur"]), list(b["depth_cur"]), role="current", config=cfg)
    d.upload_wait(ctx)
    k1 = ctx.counter("resident_launches"), ctx.counter("resident_levels")
    streamed = stats_of(ctx, cfg, streamed_refs, streamed_curs)
    assert (ctx.counter("resident_launches"), ctx.counter("resident_levels")) == (k1[0] + 1, k1[1] + 1)     # level 3 alone
    assert ctx.counter("resident_timeouts") == 0
    # (the default schedule's f16 Gram on the launch path beside the resident kernel's f32 one: a stopping rule on the edge falls
    # either way, the iteration counts are not compared here -- the fixture's variant 6 tests above do that)
    worst = 0.0
    for res in (lazy, streamed):
        for a, bb in zip(res, base):
            assert [L.Id for L in a.Statistics.Levels] == [L.Id for L in bb.Statistics.Levels]
            worst = max(worst, cm.twist_matrix_error(a.Transformation, bb.Transformation))
    print("40 pairs, coarse levels / first level resident against the launch path: twist distance at most %.1e" % worst)
    assert worst < 1e-5
    # a small streamed batch (below an eighth of the compute units) carries both flavours: the coarse levels resident, as ever
    few = 8
    d.update_raw_host_batch(streamed_curs[:few], list(b["grey_cur"][:few]), list(b["depth_cur"][:few]), role="current", config=cfg)
    d.upload_wait(ctx)
    k2 = ctx.counter("resident_levels")
    small = stats_of(ctx, cfg, streamed_refs[:few], streamed_curs[:few])
    assert ctx.counter("resident_levels") >= k2 + 2
    for a, bb in zip(small, base):
        assert cm.twist_matrix_error(a.Transformation, bb.Transformation) < 1e-5
    ctx.set_option("resident", -1)


def test_resident_edge_cases_follow_the_state_machine(ctx):
    """No constraints at all, the iteration cap, and a level that ends on its first pass: the same records as the launch path."""
    h, w = 120, 160
    I = np.random.default_rng(0).uniform(0, 255, (h, w)).astype(np.float32)
    Z = np.full((h, w), np.nan, np.float32)
    cam = d.RgbdCameraPyramid(w, h, po.FR1_K / 4, ctx)
    cam.build(3)
    a, b = cam.create(I, Z), cam.create(I, Z)
    pair = cm.synth(3, 100, 76)
    gref, gcur = gpu_pyramids(ctx, pair, 2)
    runs = {}
    for mode in (0, 1):
        ctx.set_option("resident", mode)
        runs[mode] = (stats_of(ctx, d.Config(FirstLevel=2, LastLevel=0), [a], [b])[0],
                      stats_of(ctx, d.Config(FirstLevel=1, LastLevel=0, MaxIterationsPerLevel=3, Precision=0.0), [gref], [gcur])[0],
                      stats_of(ctx, d.Config(FirstLevel=1, LastLevel=0, MaxIterationsPerLevel=1), [gref], [gcur])[0])
    empty, capped, single = runs[1]
    assert empty.isNaN() and np.allclose(empty.Transformation, np.eye(4))
    assert [L.TerminationCriterion for L in empty.Statistics.Levels] == [1, 1, 1]
    assert all(len(L.Iterations) == 1 and L.Iterations[0].ValidConstraints == 0 for L in empty.Statistics.Levels)
    for x, y in zip(runs[1], runs[0]):
        assert same_structure(x, y)
        assert np.allclose(x.Transformation, y.Transformation, atol=2e-6) and np.array_equal(np.isnan(x.Information), np.isnan(y.Information))
    assert all(len(L.Iterations) <= 3 for L in capped.Statistics.Levels) and all(len(L.Iterations) == 1 for L in single.Statistics.Levels)


def test_a_group_that_waits_in_vain_falls_back_to_the_launch_path(ctx):
    """Test hook: workgroup 1 of every group withholds its rows.  Its peers give up after a bounded number of polls, the kernel
    raises the error word, and the library repeats the batch one launch per step: same result, one time-out counted."""
    pair = cm.synth(11, 320, 240)
    gref, gcur = gpu_pyramids(ctx, pair, 3)
    cfg = d.Config(FirstLevel=2, LastLevel=0, Precision=5e-7)
    ctx.set_option("resident", 0)
    base = stats_of(ctx, cfg, [gref], [gcur])[0]
    ctx.set_option("resident", -1)
    ctx.set_option("resident_flags", 4)
    before = ctx.counter("resident_timeouts")
    res = stats_of(ctx, cfg, [gref], [gcur])[0]
    assert ctx.counter("resident_timeouts") == before + 1
    assert np.array_equal(res.Transformation, base.Transformation) and same_structure(res, base)
    ctx.set_option("resident_flags", 0)
    ctx.set_option("resident_group", 0)                                   # (the library had stopped using groups)
    again = stats_of(ctx, cfg, [gref], [gcur])[0]
    assert ctx.counter("resident_timeouts") == before + 1
    assert cm.twist_matrix_error(again.Transformation, base.Transformation) < 2e-6


def test_resident_groups_next_to_a_busy_build_stream(ctx):
    """A large frame build (build stream) is in flight while single pairs are aligned by full groups of 64 workgroups: the groups
    may have to wait for compute units, never in vain -- same records as on an idle device, no time-out, no fall-back."""
    b = datagen.synth_batch(21, 96, 640, 480)
    cam = d.RgbdCameraPyramid(640, 480, b["K"], ctx)
    cam.build(4)
    cfg = d.Config(FirstLevel=3, LastLevel=0, Precision=5e-7)
    trk = d.DenseTracker(cfg, ctx)
    refs = [cam.create_raw(b["grey_ref"][i], b["depth_ref"][i]) for i in range(4)]
    curs = [cam.create_raw(b["grey_cur"][i], b["depth_cur"][i]) for i in range(4)]
    quiet = [trk.match_batch_arrays([refs[i]], [curs[i]]) for i in range(4)]
    others = [cam.create_raw(b["grey_ref"][i], b["depth_ref"][i]) for i in range(4, 96)]
    before = ctx.counter("resident_timeouts"), ctx.counter("resident_launches")
    for rep in range(6):
        d.update_raw_host_batch(others, [b["grey_ref"][i] for i in range(4, 96)], [b["depth_ref"][i] for i in range(4, 96)], role="current", config=cfg)
        for i in range(4):
            out = trk.match_batch_arrays([refs[i]], [curs[i]])
            assert np.array_equal(out["T"], quiet[i]["T"]) and np.array_equal(out["information"], quiet[i]["information"])
    d.upload_wait(ctx)
    assert ctx.counter("resident_timeouts") == before[0] and ctx.counter("resident_launches") == before[1] + 24


def test_contexts_of_two_host_threads_run_resident_side_by_side():
    """The reference runs thread-local trackers concurrently (keyframe_graph.cpp:576-593: one validator per TBB worker).  Two host
    threads, a context each, align pairs through the resident kernel at the same time: their launches share the device (64 + 64
    workgroups), every result equals the idle device's bit for bit, no group ever times out.  (Round 2 had to serialise such launches:
    one in seven timed out -- the exchange lacked flow control towards workgroups that are idle on a coarse level.)"""
    import threading
    b = datagen.synth_batch(3, 4, 640, 480)
    cfg = d.Config(FirstLevel=3, LastLevel=1, MaxIterationsPerLevel=50, Precision=1e-4, Mu=0.05, UseInitialEstimate=True)
    guess = np.eye(4)[None]
    workers = []
    for k in range(3):
        c = d.Context(0)
        cam = d.RgbdCameraPyramid(640, 480, b["K"], c)
        cam.build(4)
        ref, cur = cam.create_raw(b["grey_ref"][k], b["depth_ref"][k]), cam.create_raw(b["grey_cur"][k], b["depth_cur"][k])
        trk = d.DenseTracker(cfg, c)
        workers.append(dict(ctx=c, trk=trk, ref=ref, cur=cur, quiet=trk.match_batch_arrays([ref], [cur], T_init=guess), bad=0))
    go = threading.Barrier(len(workers))

    def run(wk):
        go.wait()
        for _ in range(400):
            o = wk["trk"].match_batch_arrays([wk["ref"]], [wk["cur"]], T_init=guess)
            wk["bad"] += not (np.array_equal(o["T"], wk["quiet"]["T"]) and np.array_equal(o["information"], wk["quiet"]["information"]))
    ts = [threading.Thread(target=run, args=(wk,)) for wk in workers]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert sum(wk["bad"] for wk in workers) == 0
    assert sum(wk["ctx"].counter("resident_timeouts") for wk in workers) == 0
    assert all(wk["ctx"].counter("resident_launches") == 401 for wk in workers)


def test_rendezvous_of_two_threads_on_one_current_frame(ctx):
    """dvo_hip_match from two host threads with the SAME current frame -- the reference's LocalTracker (local_tracker.cpp:180-184:
    tbb::parallel_invoke of the keyframe tracker's and the odometry tracker's match): once the context has seen the two collide, the
    calls leave as one two-pair batch.  With a pinned group size a pair's record does not depend on the batch it is in: every result
    equals the single-threaded one bit for bit, statistics included; the counter shows that pairs were formed."""
    import ctypes as C
    import threading
    from dvo_slam_amd import _lib as L
    lib = ctx._lib
    b = datagen.synth_batch(21, 2, 640, 480)
    cam = d.RgbdCameraPyramid(640, 480, b["K"], ctx)
    cam.build(4)
    refs = [cam.create_raw(b["grey_ref"][i], b["depth_ref"][i]) for i in range(2)]
    cur = cam.create_raw(b["grey_cur"][0], b["depth_cur"][0])
    cfg_py = d.Config(FirstLevel=3, LastLevel=1, MaxIterationsPerLevel=50, Precision=1e-4, Mu=0.05, UseInitialEstimate=True)   # the front end's
    for f in refs + [cur]:
        f.build(cfg_py.getNumLevels())
    cfg = cfg_py.to_c()
    nl, cap = 3, 150
    ctx.set_option("variant", 7)
    ctx.set_option("resident_group", 4)

    def one(k, res, lv, it):
        for r in range(16):
            res.transformation[r] = 1.0 if r % 5 == 0 else 0.0
        return lib.dvo_hip_match(ctx.ptr, refs[k].ptr, cur.ptr, C.byref(cfg), C.byref(res), lv, nl, it, cap)

    def snapshot(res, lv, it):
        return (bytes(res), bytes(lv)[:C.sizeof(L.LevelStats) * res.n_levels], bytes(it)[:C.sizeof(L.IterationStats) * res.n_iterations_total])
    want = []
    for k in range(2):
        res, lv, it = L.Result(), (L.LevelStats * nl)(), (L.IterationStats * cap)()
        assert one(k, res, lv, it) == 0
        want.append(snapshot(res, lv, it))
    assert want[0] != want[1]
    before = ctx.counter("rendezvous_pairs")
    rounds, bad = 300, []
    go = threading.Barrier(2)

    def worker(k):
        res, lv, it = L.Result(), (L.LevelStats * nl)(), (L.IterationStats * cap)()
        for _ in range(rounds):
            go.wait()
            if one(k, res, lv, it) != 0 or snapshot(res, lv, it) != want[k]:
                bad.append(k)
    ts = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    pairs = ctx.counter("rendezvous_pairs") - before
    print("rendezvous: %d of %d concurrent call pairs left as one two-pair batch" % (pairs, rounds))
    assert not bad
    assert pairs >= 1
    ctx.set_option("rendezvous", 0)
    try:
        ts = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
        rounds = 20
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert ctx.counter("rendezvous_pairs") - before == pairs and not bad
    finally:
        ctx.set_option("rendezvous", 1)


def test_first_match_of_a_fresh_process_does_not_pay_the_wait_path_warm_up():
    """DESIGN.md section 8, "a host-side effect that looks like an engine stall": in the first GPU process on a fresh box the first stream
    wait of a batch has returned 14-24 ms after the device had finished.  dvo_hip_context_create now makes the runtime's wait path warm
    (nine waits on trivial commands, the longest kept in the counter "warmup_wait_us"); a fresh process reports the counter and the wall
    time of its very first match (code-object load included), and its second match is back at the latency of a warm process."""
    import subprocess
    import sys
    code = (
        "import sys, time, json\n"
        "sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "import dvo_slam_amd as d\n"
        "from dvo_slam_amd import datagen\n"
        "t0 = time.perf_counter(); ctx = d.Context(0); t_ctx = time.perf_counter() - t0\n"
        "pair = datagen.synth_pair(3, 640, 480)\n"
        "cam = d.RgbdCameraPyramid(640, 480, pair['K'], ctx); cam.build(4)\n"
        "ref, cur = cam.create_raw(pair['grey_ref'], pair['depth_ref']), cam.create_raw(pair['grey_cur'], pair['depth_cur'])\n"
        "trk = d.DenseTracker(d.Config(FirstLevel=3, LastLevel=0), ctx)\n"
        "ms = []\n"
        "for k in range(6):\n"
        "    r = d.Result(); t1 = time.perf_counter(); trk.match(ref, cur, r, with_stats=False); ms.append((time.perf_counter() - t1) * 1e3)\n"
        "print(json.dumps(dict(warmup_wait_us=ctx.counter('warmup_wait_us'), create_ms=t_ctx * 1e3, match_ms=ms)))\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    print("fresh process: context created in %.1f ms (longest warm-up wait %d us); matches %s ms" % (rec["create_ms"], rec["warmup_wait_us"], ["%.2f" % m for m in rec["match_ms"]]))
    assert rec["warmup_wait_us"] >= 0
    assert min(rec["match_ms"][1:]) < 1.0                      # a warm single-pair match is 0.35-0.5 ms
    assert max(rec["match_ms"][1:]) < 10.0                     # ... and none of the later ones meets a 14-24 ms wait
