#!/usr/bin/env python
"""Round 5: where the reference-compatible mode's time goes -- per-level sweep time (HIP events) and iteration counts per level, option
ref_compat 0 / 1 / 2 on the same pairs.   usage: r5_compat.py [pairs]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvo_slam_amd as d            # noqa: E402
from dvo_slam_amd import datagen    # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
b = datagen.synth_batch(0, n, 640, 480, nthreads=min(32, os.cpu_count() or 8))
for mode in (0, 1, 2):
    ctx = d.Context(0)
    ctx.set_option("ref_compat", mode)
    ctx.set_option("resident", 0)
    cam = d.RgbdCameraPyramid(640, 480, b["K"], ctx)
    cam.build(4)
    refs = [cam.create_raw(b["grey_ref"][i], b["depth_ref"][i]) for i in range(n)]
    curs = [cam.create_raw(b["grey_cur"][i], b["depth_cur"][i]) for i in range(n)]
    trk = d.DenseTracker(d.Config(FirstLevel=3, LastLevel=0), ctx)
    k_ms = [trk.time_residual_kernel(refs, curs, lvl, reps=10, warm_iterations=3) for lvl in range(4)]
    res = [d.Result() for _ in range(n)]
    trk.match_batch(refs, curs, res, with_stats=True)
    its = np.array([[len(L.Iterations) for L in r.Statistics.Levels] for r in res])          # [pair, level 3..0]
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        trk.match_batch_arrays(refs, curs)
        ts.append((time.perf_counter() - t0) * 1e3)
    print("ref_compat %d, %d pairs: sweep ms per launch, levels 0..3: %s; passes per level 3..0: mean %s, max %s; match %.3f ms"
          % (mode, n, " ".join("%.4f" % x for x in k_ms), np.round(its.mean(0), 2), its.max(0), float(np.median(ts))), flush=True)
