#!/usr/bin/env python
"""Finest-level sweep time against the tile height (option rows_per_wave) for a 512-pair launch, converged transform, weights on."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvo_slam_amd as d
from dvo_slam_amd import datagen
ctx = d.default_context()
n = 512
cam = None
refs, curs = [], []
for s in range(0, n, 128):
    b = datagen.synth_batch(s, 128, 640, 480)
    if cam is None:
        cam = d.RgbdCameraPyramid(640, 480, b["K"], ctx); cam.build(4)
    refs += [cam.create_raw(b["grey_ref"][i], b["depth_ref"][i]) for i in range(128)]
    curs += [cam.create_raw(b["grey_cur"][i], b["depth_cur"][i]) for i in range(128)]
trk = d.DenseTracker(d.Config(FirstLevel=3, LastLevel=0), ctx)
for level in (0, 1):
    for rpw in (2, 4, 8, 16):
        ctx.set_option("rows_per_wave", rpw)
        ms = [trk.time_residual_kernel(refs, curs, level, reps=10) for _ in range(3)]
        print("level %d rows_per_wave %2d: %.3f ms per %d-pair launch (%.4f per 128 pairs)" % (level, rpw, min(ms), n, min(ms) * 128 / n), flush=True)
ctx.set_option("rows_per_wave", 0)
