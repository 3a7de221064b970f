#!/usr/bin/env python
"""Round 6: the level-3 sweep of 640 x 480 pairs alone (HIP events, converged transform, weights on): the gathering sweep (small_sweep 0)
against align_small.hip (small_sweep 1) at several workgroups per pair."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvo_slam_amd as d
from dvo_slam_amd import datagen
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
b = datagen.synth_batch(0, 64, 640, 480)
ctx = d.Context(0)
ctx.set_option("resident", 0)
cam = d.RgbdCameraPyramid(640, 480, b["K"], ctx); cam.build(4)
refs = [cam.create_raw(b["grey_ref"][i % 64], b["depth_ref"][i % 64]) for i in range(n)]
curs = [cam.create_raw(b["grey_cur"][i % 64], b["depth_cur"][i % 64]) for i in range(n)]
trk = d.DenseTracker(d.Config(FirstLevel=3, LastLevel=0), ctx)
for small, tiles in ((0, 0), (1, 0), (1, 2), (1, 3), (1, 4), (1, 6), (1, 8), (0, 0), (1, 0)):
    ctx.set_option("small_sweep", small); ctx.set_option("small_tiles", tiles)
    t = min(trk.time_residual_kernel(refs, curs, 3, reps=20, warm_iterations=3) for _ in range(3))
    print("%d pairs, level 3: small_sweep %d small_tiles %d: %.1f us per launch" % (n, small, tiles, t * 1e3), flush=True)
