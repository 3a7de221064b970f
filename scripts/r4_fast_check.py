#!/usr/bin/env python
"""Round 4: the contracted window sweep (variants 8 / 9, align_fast.hip) against the bit-exact one (variant 7): residual and constraint
differences of one linearisation, whole matches, then timings of the finest-level launch.  Usage: r4_fast_check.py [variants] [--no-timing]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvo_slam_amd as d            # noqa: E402
from dvo_slam_amd import datagen    # noqa: E402
from oracle import pyoracle as po   # noqa: E402

variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "8,9").split(",")]
big = int(os.environ.get("DVO_BIG", "1024"))


def ctx_with(variant, rpw=0):
    ctx = d.Context(0)
    ctx.set_option("variant", variant)
    ctx.set_option("resident", 0)
    if rpw:
        ctx.set_option("rows_per_wave", rpw)
    return ctx


def frames(ctx, pair, w, h, levels):
    cam = d.RgbdCameraPyramid(w, h, pair["K"], ctx)
    cam.build(levels)
    return cam.create_raw(pair["grey_ref"], pair["depth_ref"]), cam.create_raw(pair["grey_cur"], pair["depth_cur"])


for seed, (w, h), xi in ((11, (640, 480), [0.004, -0.003, 0.002, 0.006, -0.004, 0.003]), (12, (320, 240), [0.02, 0.01, -0.015, -0.02, 0.025, 0.03]),
                         (13, (640, 480), [0.05, -0.04, 0.03, 0.05, 0.04, -0.06]), (14, (128, 96), [0.0, 0, 0, 0, 0, 0]),
                         (15, (640, 480), [0.3, -0.2, 0.1, 0.2, 0.3, -0.4]), (16, (1280, 960), [0.01, -0.01, 0.005, 0.01, 0.01, -0.01])):
    pair = datagen.synth_pair(seed, w, h)
    T34 = po.se3_exp(np.array(xi))[:3]
    out = {}
    for v in [7] + variants:
        ctx = ctx_with(v, 4)
        ref, cur = frames(ctx, pair, w, h, 1)
        trk = d.DenseTracker(d.Config(FirstLevel=0, LastLevel=0), ctx)
        out[v] = [trk.level_iteration(ref, cur, 0, T34, P_prev=[900.0, 3.0, 3.0, 400.0], first=f, want_residuals=True) for f in (True, False)]
        out[v].append(ctx.counter("window_fallbacks"))
    for v in variants:
        for k in (0, 1):
            a, b = out[7][k], out[v][k]
            ra, rb = a["residuals"].reshape(-1, 2), b["residuals"].reshape(-1, 2)
            va, vb = ~np.isnan(ra[:, 0]), ~np.isnan(rb[:, 0])
            both = va & vb
            d0 = np.abs(ra[both, 0] - rb[both, 0]).max() if both.any() else 0.0
            d1 = np.abs(ra[both, 1] - rb[both, 1]).max() if both.any() else 0.0
            dA = np.abs(a["A"] - b["A"]).max() / np.abs(a["A"]).max()
            db = np.abs(a["b"] - b["b"]).max() / np.abs(a["b"]).max()
            print("seed %d %dx%d |xi| %.2f first=%d variant %d: n %d vs %d (only in 7: %d, only in %d: %d), max |dr0| %.2e |dr1| %.2e, A rel %.1e, b rel %.1e, -ll %.9g vs %.9g, fallbacks %d"
                  % (seed, w, h, np.abs(xi).max(), 1 - k, v, a["n"], b["n"], int((va & ~vb).sum()), v, int((vb & ~va).sum()), d0, d1, dA, db,
                     a["neg_ll"], b["neg_ll"], out[v][2]), flush=True)

pair = datagen.synth_pair(1234, 640, 480)
res = {}
for v in [7] + variants:
    ctx = ctx_with(v, 4)
    ref, cur = frames(ctx, pair, 640, 480, 4)
    r = d.Result()
    d.DenseTracker(d.Config(FirstLevel=3, LastLevel=0), ctx).match(ref, cur, r)
    res[v] = r
    print("variant %d: iterations %s" % (v, [(L.Id, len(L.Iterations)) for L in r.Statistics.Levels]))
for v in variants:
    print("variant %d vs 7: max |dT| %.2e" % (v, np.abs(res[7].Transformation - res[v].Transformation).max()))

if "--no-timing" not in sys.argv:
    for n in (128, big):
        b = datagen.synth_batch(0, min(n, 128), 640, 480)
        for v in [7] + variants:
            ctx = ctx_with(v)
            cam = d.RgbdCameraPyramid(640, 480, b["K"], ctx)
            cam.build(4)
            m = min(n, 128)
            refs = [cam.create_raw(b["grey_ref"][i % m], b["depth_ref"][i % m]) for i in range(n)]
            curs = [cam.create_raw(b["grey_cur"][i % m], b["depth_cur"][i % m]) for i in range(n)]
            trk = d.DenseTracker(d.Config(FirstLevel=3, LastLevel=0), ctx)
            for level in (0, 1):
                ms = [trk.time_residual_kernel(refs, curs, level, reps=10, warm_iterations=3) for _ in range(3)]
                px = (640 >> level) * (480 >> level)
                print("pairs %d level %d variant %d: %s ms -> %.0f GB/s at 40 B/px = %.3f of 8 TB/s; fallbacks %d"
                      % (n, level, v, ["%.4f" % x for x in ms], 40.0 * px * n / min(ms) / 1e6, 40.0 * px * n / min(ms) / 1e6 / 8000.0, ctx.counter("window_fallbacks")), flush=True)
            del refs, curs, trk, cam, ctx
