#!/usr/bin/env python
"""Where the time of a sweep's tail goes (DVO_TAIL_CLOCKS build of the library, scripts/ubench/_build/tailclk): the gathering sweep of
level 3 with the wide half of the solver step in its tail (option sweep_tail 2), n pairs of 640 x 480.  Phases, 100 MHz wall clock of
thread 0: 0 tile | 1 arrival (stores' completion, barriers, ticket) | 2 arguments staged | 3 reduction | 4 log-likelihood | 5 sums
written | 6 the wide half as its caller sees it."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dvo_slam_amd as d
from dvo_slam_amd import datagen

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
W, H = 640, 480
b = datagen.synth_batch(0, 8, W, H)
ctx = d.Context(0)
cam = d.RgbdCameraPyramid(W, H, b["K"], ctx); cam.build(4)
fr = [cam.create_raw(b["grey_ref"][i], b["depth_ref"][i]) for i in range(8)]
fc = [cam.create_raw(b["grey_cur"][i], b["depth_cur"][i]) for i in range(8)]
refs, curs = [fr[i % 8] for i in range(n)], [fc[i % 8] for i in range(n)]
trk = d.DenseTracker(d.Config(FirstLevel=3, LastLevel=3), ctx)     # level 3 alone: the gathering sweep
lib = ctx._lib
out = (C.c_ulonglong * 16)()
for mode in (2,):
    ctx.set_option("resident", 0)
    ctx.set_option("sweep_tail", mode)
    trk.match_batch_arrays(refs, curs)
    lib.dvo_hip_debug_tail_clocks(out, 1)
    for _ in range(5):
        trk.match_batch_arrays(refs, curs)
    lib.dvo_hip_debug_tail_clocks(out, 1)
    names = ["tile", "arrival", "args staged", "reduction", "log-likelihood", "sums written", "wide half (caller)"]
    for k, name in enumerate(names):
        cnt = out[8 + k]
        print("%d pairs, sweep_tail %d: %-20s %8.2f us average over %d" % (n, mode, name, (out[k] / cnt / 100.0) if cnt else 0.0, cnt))
