#!/usr/bin/env python
"""Iteration counts per level over many pairs for the schedules of the sweep: a noise floor of the normal equations close to the
stopping precision shows as a tail of long levels (termination by "increment too small" missed, the level runs until the
log-likelihood stops improving).  usage: iteration_tails.py [pairs] [variants]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvo_slam_amd as d            # noqa: E402
from dvo_slam_amd import datagen, parallel    # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
variants = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "5,7").split(",")]
b = datagen.synth_batch(0, n, 640, 480)
for v in variants:
    ctx = d.Context(0)
    ctx.set_option("variant", v)
    ctx.set_option("resident", 0)
    cam = d.RgbdCameraPyramid(640, 480, b["K"], ctx)
    cam.build(4)
    refs = [cam.create_raw(b["grey_ref"][i], b["depth_ref"][i]) for i in range(n)]
    curs = [cam.create_raw(b["grey_cur"][i], b["depth_cur"][i]) for i in range(n)]
    trk = d.DenseTracker(d.Config(FirstLevel=3, LastLevel=0), ctx)
    results = [d.Result() for _ in range(n)]
    trk.match_batch(refs, curs, results, with_stats=True)
    its = np.array([[len(L.Iterations) for L in r.Statistics.Levels] for r in results])
    err = np.array([np.abs(parallel.twists_of(r.Transformation[None])[0] - b["xi_true"][i]).max() for i, r in enumerate(results)])
    print("variant %d, %d pairs: iterations per level (3..0) mean %s, max %s, 99th percentile %s; pairs with a level longer than 20 iterations: %d; "
          "distance to the true motion median %.2e max %.2e" % (v, n, np.round(its.mean(0), 2), its.max(0), np.percentile(its, 99, axis=0), int((its > 20).any(1).sum()),
                                                            np.median(err), err.max()), flush=True)
    del refs, curs, trk, cam, ctx
