#!/usr/bin/env python
"""Experiment driver: one configuration per process (env DVO_EXP_*), prints the finest-level sweep time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvo_slam_amd as d
from dvo_slam_amd import datagen
N = 128
ctx = d.default_context()
b = datagen.synth_batch(0, N, 640, 480)
cam = d.RgbdCameraPyramid(640, 480, b["K"], ctx)
cam.build(4)
refs = [cam.create_raw(b["grey_ref"][i], b["depth_ref"][i]) for i in range(N)]
curs = [cam.create_raw(b["grey_cur"][i], b["depth_cur"][i]) for i in range(N)]
trk = d.DenseTracker(d.Config(FirstLevel=3, LastLevel=0), ctx)
ctx.set_option("inkernel_ll", int(os.environ.get("DVO_INKERNEL_LL", "0")))
ctx.set_option("rows_per_wave", int(os.environ.get("DVO_ROWS_PER_WAVE", "8")))
ms = [trk.time_residual_kernel(refs, curs, 0, reps=20, warm_iterations=3) for _ in range(3)]
print("skip=%s mode3=%s lds=%s rpw=%s: %.4f ms (%s)" % (os.environ.get("DVO_EXP_SKIP", "0"), os.environ.get("DVO_EXP_MODE3", "0"), os.environ.get("DVO_EXP_LDS", "0"),
      os.environ.get("DVO_ROWS_PER_WAVE", "8"), min(ms), " ".join("%.4f" % m for m in ms)), flush=True)
