#!/usr/bin/env python
"""Time per match of the tracking front end's configuration (dvo_benchmark/launch/benchmark.yaml: levels 3..1, Precision 1e-4, Mu 0.05,
initial estimate) for 1 and 2 pairs, resident kernel against the launch path; termination criteria of the levels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dvo_slam_amd as d
from dvo_slam_amd import datagen
ctx = d.default_context()
seq = datagen.synth_sequence(5, 12, 640, 480, depth_noise=1.0, grey_noise=2.0, exposure=0.0)
cam = d.RgbdCameraPyramid(640, 480, seq["K"], ctx); cam.build(4)
frames = [cam.create_raw(seq["grey"][k], seq["depth"][k]) for k in range(12)]
cfg = d.Config(FirstLevel=3, LastLevel=1, MaxIterationsPerLevel=50, Precision=1e-4, Mu=0.05, UseInitialEstimate=True)
trk = d.DenseTracker(cfg, ctx)
rel = [np.linalg.inv(seq["poses"][k]) @ seq["poses"][k + 1] for k in range(11)]
for mode in (0, -1):
    ctx.set_option("resident", mode)
    for n in (1, 2):
        T0 = np.stack([np.linalg.inv(rel[3])] * n)      # previous motion as the guess, like the replay loop does
        refs, curs = [frames[4]] * n, [frames[5]] * n
        for _ in range(5):
            out = trk.match_batch_arrays(refs, curs, T_init=T0)
        t0 = time.perf_counter()
        for _ in range(100):
            out = trk.match_batch_arrays(refs, curs, T_init=T0)
        dt = (time.perf_counter() - t0) / 100 * 1e3
        r = d.Result(); r.Transformation = T0[0].copy()
        trk.match_batch([frames[4]], [frames[5]], [r], with_stats=True)
        print("resident=%2d pairs %d: %.3f ms per match; iterations %s; levels (id, iterations, termination) %s"
              % (mode, n, dt, out["n_iterations"][:2], [(L.Id, len(L.Iterations), L.TerminationCriterion) for L in r.Statistics.Levels]), flush=True)
