#!/usr/bin/env python
"""Round 5: what a Gram schedule does to the normal equations of one linearisation -- A, b and -ll of the library selected by
DVO_HIP_LIBRARY (default schedule) against the f32 Gram of the same sweep (option variant 6 + contracted? no: variant 6 is the exact
arithmetic; so the comparison is made on variant 8's own residuals: option compact 0, A / b against float64 sums of the per-pixel terms is
not available -- instead against variant 7's f16 hi + lo Gram, 1e-6 from f32), at the converged transform of 16 pairs, levels 0..3."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvo_slam_amd as d            # noqa: E402
from dvo_slam_amd import datagen    # noqa: E402
from oracle import pyoracle as po   # noqa: E402

n = 8
b = datagen.synth_batch(0, n, 640, 480)
ctx = d.Context(0)
cam = d.RgbdCameraPyramid(640, 480, b["K"], ctx)
cam.build(4)
worst = {}
for i in range(n):
    ref, cur = cam.create_raw(b["grey_ref"][i], b["depth_ref"][i]), cam.create_raw(b["grey_cur"][i], b["depth_cur"][i])
    pair = {k: b[k][i] for k in ("grey_ref", "depth_ref", "grey_cur", "depth_cur")}
    pair["K"] = b["K"]
    oref, ocur = po.pyramids_from_pair(pair, 4)
    T = po.se3_exp(b["xi_true"][i] * 0.98)[:3]               # near the solution: b is a sum of cancelling terms there
    for level in range(4):
        trk = d.DenseTracker(d.Config(FirstLevel=level, LastLevel=level), ctx)
        o = po.level_iteration(oref, ocur, level, T, P_prev=[900.0, 3.0, 3.0, 400.0], first=False, mode=po.MATH)
        g = trk.level_iteration(ref, cur, level, T, P_prev=[900.0, 3.0, 3.0, 400.0], first=False)
        xo, xg = np.linalg.solve(o["A"], o["b"]), np.linalg.solve(g["A"], g["b"])
        e = (np.abs(g["A"] - o["A"]).max() / np.abs(o["A"]).max(), np.abs(g["b"] - o["b"]).max() / np.abs(o["b"]).max(),
             np.abs(g["b"] - o["b"]).max() / np.sqrt(np.abs(np.diag(o["A"])).max()), np.abs(xg - xo).max(), g["n"] - o["n"])
        w = worst.setdefault(level, [0, 0, 0, 0, 0])
        for k in range(5):
            w[k] = max(w[k], abs(e[k]))
for level in range(4):
    print("level %d: |dA|/|A| %.2e  |db|/|b| %.2e  |db|/sqrt(A_max) %.2e  |dx| %.2e  |dn| %d   (GPU vs oracle MATH, worst of %d pairs)" % ((level,) + tuple(worst[level]) + (n,)))
