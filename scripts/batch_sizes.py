#!/usr/bin/env python
"""Match time against batch size (1 .. 256 pairs of 640x480) for the three configurations of the reference's callers: the validator's
screening stage (level 3 only), its refinement / the front end (levels 3 -> 1, initial estimate, Mu 0.05) and the full BASELINE match
(levels 3 -> 0): median ms per dvo_hip_match_batch, which path ran (resident launches), where the host thread's time went."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvo_slam_amd as d            # noqa: E402
from dvo_slam_amd import datagen    # noqa: E402

sizes = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "1,2,4,8,16,32,64,128,256".split(","))]
b = datagen.synth_batch(7, 64, 640, 480)
ctx = d.Context(0)
if os.environ.get("DVO_RESIDENT"):
    ctx.set_option("resident", int(os.environ["DVO_RESIDENT"]))       # 0: every level on the launch path
cam = d.RgbdCameraPyramid(640, 480, b["K"], ctx)
cam.build(4)
nmax = max(sizes)
refs = [cam.create_raw(b["grey_ref"][i % 64], b["depth_ref"][i % 64]) for i in range(nmax)]
curs = [cam.create_raw(b["grey_cur"][i % 64], b["depth_cur"][i % 64]) for i in range(nmax)]
configs = [("screen L3", d.Config(FirstLevel=3, LastLevel=3, MaxIterationsPerLevel=100, Precision=1e-4, Mu=0.05, UseInitialEstimate=True)),
           ("refine 3->1", d.Config(FirstLevel=3, LastLevel=1, MaxIterationsPerLevel=100, Precision=1e-4, Mu=0.05, UseInitialEstimate=True)),
           ("full 3->0", d.Config(FirstLevel=3, LastLevel=0))]
keys = ("host_ns_prepare", "host_ns_enqueue", "host_ns_wait", "host_ns_finish")
for name, cfg in configs:
    trk = d.DenseTracker(cfg, ctx)
    for n in sizes:
        guess = np.stack([np.eye(4)] * n)
        trk.match_batch_arrays(refs[:n], curs[:n], T_init=guess)
        r0 = ctx.counter("resident_launches")
        h0 = [ctx.counter(k) for k in keys]
        ts = []
        reps = 30 if n <= 64 else 12
        for _ in range(reps):
            t0 = time.perf_counter()
            out = trk.match_batch_arrays(refs[:n], curs[:n], T_init=guess)
            ts.append((time.perf_counter() - t0) * 1e3)
        h1 = [ctx.counter(k) for k in keys]
        host = [(a - b0) / reps / 1e6 for a, b0 in zip(h1, h0)]
        print("%-12s %4d pairs: median %.3f ms (min %.3f), %.4f ms per pair; resident launches per match %.1f; host thread: prepare %.3f enqueue %.3f wait %.3f finish %.3f ms; iterations %.1f"
              % (name, n, np.median(ts), min(ts), np.median(ts) / n, (ctx.counter("resident_launches") - r0) / reps, host[0], host[1], host[2], host[3],
                 out["n_iterations"].mean()), flush=True)
