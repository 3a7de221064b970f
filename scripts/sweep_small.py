#!/usr/bin/env python
"""Kernel-time sweep of the reduce kernel over tile heights for every level (default variant)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvo_slam_amd as d            # noqa: E402
from dvo_slam_amd import datagen    # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ctx = d.default_context()
b = datagen.synth_batch(0, n, 640, 480)
cam = d.RgbdCameraPyramid(640, 480, b["K"], ctx)
cam.build(4)
refs = [cam.create_raw(b["grey_ref"][i], b["depth_ref"][i]) for i in range(n)]
curs = [cam.create_raw(b["grey_cur"][i], b["depth_cur"][i]) for i in range(n)]
trk = d.DenseTracker(d.Config(FirstLevel=3, LastLevel=0), ctx)
for m in (n, 1):
    for level in (3, 2, 1, 0):
        line = []
        for rpw in (1, 2, 4, 8, 16):
            ctx.set_option("rows_per_wave", rpw)
            line.append("rpw%2d %.4f" % (rpw, trk.time_residual_kernel(refs[:m], curs[:m], level, reps=20)))
        print("pairs=%d level=%d  " % (m, level) + "  ".join(line), flush=True)
