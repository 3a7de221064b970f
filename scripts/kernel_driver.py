#!/usr/bin/env python
"""Launches the finest-level reduce kernel a few times on a 128-pair batch (for rocprofv3 --pmc passes)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvo_slam_amd as d            # noqa: E402
from dvo_slam_amd import datagen    # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
level = int(sys.argv[2]) if len(sys.argv) > 2 else 0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
ctx = d.default_context()
for k in ("variant", "rows_per_wave"):
    if os.environ.get("DVO_" + k.upper()):
        ctx.set_option(k, int(os.environ["DVO_" + k.upper()]))
b = datagen.synth_batch(0, n, 640, 480)
cam = d.RgbdCameraPyramid(640, 480, b["K"], ctx)
cam.build(4)
refs = [cam.create_raw(b["grey_ref"][i], b["depth_ref"][i]) for i in range(n)]
curs = [cam.create_raw(b["grey_cur"][i], b["depth_cur"][i]) for i in range(n)]
trk = d.DenseTracker(d.Config(FirstLevel=3, LastLevel=0), ctx)
warm = int(os.environ.get("DVO_WARM", "3"))
print("avg ms", trk.time_residual_kernel(refs, curs, level, reps=reps, warm_iterations=warm))
