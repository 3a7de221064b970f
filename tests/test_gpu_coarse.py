"""The fused coarse-level kernel (dvo_slam_amd/csrc/align_coarse.hip: ONE launch runs the coarse pyramid levels of every pair of a batch,
a workgroup per pair, from gn_level_begin to the level's termination) against the launch-per-step path of the same library.

Both paths are made of the same device functions on the same data layout -- the tile sweeps (fast_sweep.h, mfma_sweep.h), stage 3 of the
reduction, the fused log-likelihood, the loop body (solver_logic.h: dense_tracking.cpp:200-357) -- so the bar is BIT identity: every byte
of every result, level record and iteration record.  The reference runs one match() per thread from start to finish
(dvo_slam/src/keyframe_graph.cpp:576-593); this is the kernel that does the same with a workgroup."""
import ctypes as C

import numpy as np
import pytest

import common as cm
import dvo_slam_amd as d
from dvo_slam_amd import _lib, datagen
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

@pytest.fixture()
def ctx():
    c = d.Context(0)
    yield c


def raw_match(ctx, cfg, refs, curs, T0=None):
    """dvo_hip_match_batch with statistics; returns the raw bytes of the three output arrays + the results as a structured view"""
    n = len(refs)
    for r, c in zip(refs, curs):
        r.build(cfg.getNumLevels())
        c.build(cfg.getNumLevels())
    cres = (_lib.Result * n)()
    for i in range(n):
        T = np.eye(4) if T0 is None else np.asarray(T0[i], np.float64)
        for k, v in enumerate(T.reshape(-1)):
            cres[i].transformation[k] = v
    vp = C.c_void_p
    rp = (vp * n)(*[p.ptr for p in refs])
    cp = (vp * n)(*[p.ptr for p in curs])
    nl = cfg.FirstLevel - cfg.LastLevel + 1
    cap = nl * cfg.MaxIterationsPerLevel
    levels = (_lib.LevelStats * (n * nl))()
    iters = (_lib.IterationStats * (n * cap))()
    ccfg = cfg.to_c()
    ctx.check(ctx._lib.dvo_hip_match_batch(ctx.ptr, n, rp, cp, C.byref(ccfg), cres, levels, nl, iters, cap))
    return bytes(cres), bytes(levels), bytes(iters), cres


def frames_of(ctx, batch, w, h, levels, n):
    cam = d.RgbdCameraPyramid(w, h, batch["K"], ctx)
    cam.build(levels)
    refs = [cam.create_raw(batch["grey_ref"][i], batch["depth_ref"][i]) for i in range(n)]
    curs = [cam.create_raw(batch["grey_cur"][i], batch["depth_cur"][i]) for i in range(n)]
    return refs, curs


@pytest.mark.parametrize("w,h,first,last,mu,init,precision,n,fused_levels", [
    (640, 480, 3, 0, 0.0, False, 5e-7, 5, 2),   # BASELINE config 4's shape: levels 3 (gathered taps) and 2 (window sweep, half-empty tile column) fused
    (640, 480, 3, 2, 0.0, False, 5e-7, 3, 2),   # the whole match inside the kernel: it writes the results (gn_finish)
    (640, 480, 3, 1, 0.05, True, 1e-4, 4, 2),   # benchmark.yaml: motion prior, initial estimate
    (320, 240, 3, 0, 0.0, False, 5e-7, 6, 3),   # 40 x 30, 80 x 60 (gathered) and 160 x 120 (window)
    (262, 194, 2, 0, 0.0, False, 5e-7, 3, 2),   # ragged: 65 x 48 and 131 x 97 (odd widths: gathered, pixels in linear order)
    (1280, 960, 4, 0, 0.0, False, 1e-4, 2, 2),  # BASELINE config 5: 80 x 60 and 160 x 120 again, under five levels
    (256, 192, 2, 0, 0.0, False, 5e-7, 3, 0),   # 64 x 48: full tiles below the contracted sweep's 84 columns go to the exact window sweep, which
                                                # the fused kernel has no instantiation for -- it declines, the launch path runs every level
])
def test_coarse_kernel_records_are_the_launch_path_s_bit_for_bit(ctx, w, h, first, last, mu, init, precision, n, fused_levels):
    b = datagen.synth_batch(300 + w, n, w, h)
    refs, curs = frames_of(ctx, b, w, h, first + 1, n)
    cfg = d.Config(FirstLevel=first, LastLevel=last, Mu=mu, UseInitialEstimate=init, Precision=precision, MaxIterationsPerLevel=50 if init else 100)
    T0 = [po.se3_exp(0.5 * np.asarray(b["xi_true"][i])) for i in range(n)] if init else None
    ctx.set_option("resident", 0)
    ctx.set_option("small_sweep", 0)                   # (the fused kernel has the gathering sweep for small levels, not align_small.hip's)
    ctx.set_option("rows_per_wave", 2)                 # the fused kernel's tile height on the levels that gather their taps
    ctx.set_option("coarse", 0)
    before, levels_before = ctx.counter("coarse_launches"), ctx.counter("coarse_levels")
    base = raw_match(ctx, cfg, refs, curs, T0)
    assert ctx.counter("coarse_launches") == before
    ctx.set_option("coarse", 1)
    fused = raw_match(ctx, cfg, refs, curs, T0)
    assert ctx.counter("coarse_launches") == before + (1 if fused_levels else 0)
    assert ctx.counter("coarse_levels") == levels_before + fused_levels
    assert fused[0] == base[0], "results differ"
    assert fused[1] == base[1], "level records differ"
    assert fused[2] == base[2], "iteration records differ"
    # three workgroups per compute unit (168 registers): the same bits
    ctx.set_option("coarse_workgroups", 3)
    three = raw_match(ctx, cfg, refs, curs, T0)
    assert three[:3] == base[:3]
    # ... and it is a real alignment: the oracle's transform
    if not init and last == 0:
        pair = {k: b[k][0] for k in ("grey_ref", "depth_ref", "grey_cur", "depth_cur")}
        pair["K"] = b["K"]
        oref, ocur = po.pyramids_from_pair(pair, first + 1)
        o = po.match(oref, ocur, po.make_config(first, last, 100, precision, mode=po.MATH))
        T = np.array(fused[3][0].transformation).reshape(4, 4)
        assert cm.twist_matrix_error(T, o["T"]) < (2e-5 if precision > 1e-6 else 2e-6)


def test_the_fused_kernel_is_opt_in_and_declines_what_it_has_no_instantiation_for(ctx):
    """Default: off (measured slower than the launch path up to 1024 pairs per batch).  Switched on, it takes the
    default schedule only: under "deterministic" (f32 Gram, the exact window sweep) and "ref_compat" every level stays on the launch path."""
    w, h, n = 320, 240, 5
    b = datagen.synth_batch(77, n, w, h)
    refs, curs = frames_of(ctx, b, w, h, 4, n)
    trk = d.DenseTracker(d.Config(FirstLevel=3, LastLevel=0), ctx)
    order = [i % n for i in range(160)]
    before = ctx.counter("coarse_launches")
    ctx.set_option("small_sweep", 0)
    base = trk.match_batch_arrays([refs[i] for i in order], [curs[i] for i in order])
    assert ctx.counter("coarse_launches") == before
    ctx.set_option("coarse", 1)
    fused = trk.match_batch_arrays([refs[i] for i in order], [curs[i] for i in order])
    assert ctx.counter("coarse_launches") == before + 1
    for k in range(len(order)):
        assert cm.twist_matrix_error(fused["T"][k], base["T"][k]) < 2e-6
    # every copy of a pair gets the same bytes (a workgroup's arithmetic does not depend on its neighbours)
    for k, i in enumerate(order):
        assert np.array_equal(fused["T"][k], fused["T"][i]) and fused["n_iterations"][k] == fused["n_iterations"][i]
    for key in ("deterministic", "ref_compat"):
        ctx.set_option(key, 1)
        out = trk.match_batch_arrays(refs, curs)
        assert ctx.counter("coarse_launches") == before + 1
        assert max(cm.twist_matrix_error(out["T"][k], base["T"][k]) for k in range(n)) < (1e-4 if key == "ref_compat" else 2e-6)
        ctx.set_option(key, 0)


def test_pairs_leave_the_fused_kernel_independently(ctx):
    """A batch whose pairs need very different iteration counts -- an identical pair (leaves every level after its first passes), pairs
    with holes, a pair without a single selected pixel: each one's record is what it gets alone."""
    w, h = 640, 480
    b = datagen.synth_batch(900, 3, w, h)
    cam = d.RgbdCameraPyramid(w, h, b["K"], ctx)
    cam.build(4)
    flat = np.full((h, w), 128, np.uint8)                                # no gradient: nothing is selected
    frames_r = [cam.create_raw(b["grey_ref"][0], b["depth_ref"][0]), cam.create_raw(b["grey_ref"][1], b["depth_ref"][1]),
                cam.create_raw(flat, b["depth_ref"][2]), cam.create_raw(b["grey_ref"][2], b["depth_ref"][2])]
    frames_c = [cam.create_raw(b["grey_ref"][0], b["depth_ref"][0]), cam.create_raw(b["grey_cur"][1], b["depth_cur"][1]),
                cam.create_raw(flat, b["depth_cur"][2]), cam.create_raw(b["grey_cur"][2], b["depth_cur"][2])]
    cfg = d.Config(FirstLevel=3, LastLevel=0)
    ctx.set_option("resident", 0)
    ctx.set_option("small_sweep", 0)
    ctx.set_option("rows_per_wave", 2)
    ctx.set_option("coarse", 1)
    together = raw_match(ctx, cfg, frames_r, frames_c)
    size = C.sizeof(_lib.Result)
    for i in range(4):
        alone = raw_match(ctx, cfg, frames_r[i:i + 1], frames_c[i:i + 1])
        assert alone[0] == together[0][i * size:(i + 1) * size]
    ctx.set_option("coarse", 0)
    launched = raw_match(ctx, cfg, frames_r, frames_c)
    assert launched[:3] == together[:3]
