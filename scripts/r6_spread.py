#!/usr/bin/env python
"""Round 6: the slow lane (option "overlap_tails") on a batch whose pairs differ widely in the passes they need -- pairs (i, i + k) of one
camera sweep with baselines k of very different length, what a loop-closure validator's proposals look like -- against the synchronous chain.
Match only (frames built once), options alternated.   python scripts/r6_spread.py [pairs] [fraction ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import dvo_slam_amd as d
from dvo_slam_amd import datagen

only = None                                                             # --only sync | lane: one mode, for a profiler run
if "--only" in sys.argv:
    k = sys.argv.index("--only")
    only = sys.argv[k + 1]
    del sys.argv[k:k + 2]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
fractions = [int(a) for a in sys.argv[2:]] or [8, 4]
W, H, F = 640, 480, 48
seq = datagen.synth_sequence(7, F, W, H)
ctx = d.Context(0)
cam = d.RgbdCameraPyramid(W, H, seq["K"], ctx)
cam.build(4)
frames = [cam.create_raw(seq["grey"][i], seq["depth"][i]) for i in range(F)]
rng = np.random.default_rng(3)
baselines = rng.choice([1, 1, 1, 1, 1, 1, 1, 2, 2, 3, 5, 8], size=n)
starts = rng.integers(0, F - 9, size=n)
refs = [frames[s] for s in starts]
curs = [frames[s + k] for s, k in zip(starts, baselines)]
trk = d.DenseTracker(d.Config(FirstLevel=3, LastLevel=0), ctx)


def timed(reps=6):
    trk.match_batch_arrays(refs, curs)
    t0 = time.perf_counter()
    for _ in range(reps):
        out = trk.match_batch_arrays(refs, curs)
    return (time.perf_counter() - t0) / reps * 1e3, out


ms, base = timed(2)
its = np.asarray(base["n_iterations"])
print("pairs %d; passes per pair: min %d, median %d, 90th percentile %d, max %d" % (n, its.min(), np.median(its), np.percentile(its, 90), its.max()))
if only:
    ctx.set_option("overlap_tails", 1 if only == "lane" else 0)
    ctx.set_option("overlap_fraction", fractions[0])
    print(only, "%.3f ms" % timed(4)[0])
    sys.exit(0)
for rep in range(3):
    ctx.set_option("overlap_tails", 0)
    ms0, _ = timed()
    line = "synchronous chain %.3f ms" % ms0
    for f in fractions:
        ctx.set_option("overlap_tails", 1)
        ctx.set_option("overlap_fraction", f)
        ms1, out = timed()
        same = all(np.array_equal(out["T"][k], base["T"][k]) for k in range(n))
        line += "   lane at 1/%d: %.3f ms (%s)" % (f, ms1, "same bits" if same else "DIFFERENT")
    print(line, " tails", ctx.counter("overlapped_tails"), "wait us", ctx.counter("tail_wait_us"))
