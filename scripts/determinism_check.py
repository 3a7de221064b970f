#!/usr/bin/env python
"""Run-to-run determinism of the match: the same batch 300 times, every result compared bitwise with the first (1, 2, 8, 40 pairs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dvo_slam_amd as d
from dvo_slam_amd import datagen
ctx = d.default_context()
for (w, h, lv) in ((640, 480, 4), (320, 240, 3)):
    b = datagen.synth_batch(5, 40, w, h)
    cam = d.RgbdCameraPyramid(w, h, b["K"], ctx); cam.build(lv)
    refs = [cam.create_raw(b["grey_ref"][i], b["depth_ref"][i]) for i in range(40)]
    curs = [cam.create_raw(b["grey_cur"][i], b["depth_cur"][i]) for i in range(40)]
    trk = d.DenseTracker(d.Config(FirstLevel=lv - 1, LastLevel=0, MaxIterationsPerLevel=100, Precision=5e-7), ctx)
    for n in (1, 2, 8, 40):
        first = trk.match_batch_arrays(refs[:n], curs[:n])
        bad = 0
        for rep in range(300):
            out = trk.match_batch_arrays(refs[:n], curs[:n])
            if not (np.array_equal(out["T"], first["T"]) and np.array_equal(out["information"], first["information"], equal_nan=True) and np.array_equal(out["n_iterations"], first["n_iterations"])):
                bad += 1
        print(w, h, "pairs", n, "runs that differ from the first:", bad, "of 300", flush=True)
