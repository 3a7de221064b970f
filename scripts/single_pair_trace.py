#!/usr/bin/env python
"""One 640x480 pair aligned a few times (for a rocprofv3 --kernel-trace timeline of the latency case)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvo_slam_amd as d
from dvo_slam_amd import datagen
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ctx = d.default_context()
if os.environ.get("DVO_RESIDENT"):
    ctx.set_option("resident", int(os.environ["DVO_RESIDENT"]))
b = datagen.synth_batch(0, n, 640, 480)
cam = d.RgbdCameraPyramid(640, 480, b["K"], ctx); cam.build(4)
refs = [cam.create_raw(b["grey_ref"][i], b["depth_ref"][i]) for i in range(n)]
curs = [cam.create_raw(b["grey_cur"][i], b["depth_cur"][i]) for i in range(n)]
trk = d.DenseTracker(d.Config(FirstLevel=3, LastLevel=0, MaxIterationsPerLevel=100, Precision=5e-7, Mu=0.0), ctx)
import time
for _ in range(6):
    out = trk.match_batch_arrays(refs, curs)
    time.sleep(0.002)
print(out["n_iterations"])
