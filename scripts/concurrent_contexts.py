#!/usr/bin/env python
"""Two (or more) host threads, one context each, aligning single pairs / pairs of pairs concurrently on ONE GPU through the resident
kernel -- the reference's thread-local trackers (dvo_slam/src/keyframe_graph.cpp:576-593, local_tracker.cpp:180-184).  Counts
time-outs of workgroup groups, compares every result with the one an idle device gives.
usage: concurrent_contexts.py [threads] [matches per thread] [pairs per match]"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvo_slam_amd as d            # noqa: E402
from dvo_slam_amd import datagen    # noqa: E402

threads = int(sys.argv[1]) if len(sys.argv) > 1 else 2
per_thread = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
pairs = int(sys.argv[3]) if len(sys.argv) > 3 else 1
b = datagen.synth_batch(3, 4, 640, 480)
cfg = d.Config(FirstLevel=3, LastLevel=1, MaxIterationsPerLevel=50, Precision=1e-4, Mu=0.05, UseInitialEstimate=True)    # the front end's configuration
guess = np.stack([np.eye(4)] * pairs)


def worker(k, out):
    ctx = d.Context(0)
    cam = d.RgbdCameraPyramid(640, 480, b["K"], ctx)
    cam.build(4)
    refs = [cam.create_raw(b["grey_ref"][(k + i) % 4], b["depth_ref"][(k + i) % 4]) for i in range(pairs)]
    curs = [cam.create_raw(b["grey_cur"][(k + i) % 4], b["depth_cur"][(k + i) % 4]) for i in range(pairs)]
    trk = d.DenseTracker(cfg, ctx)
    quiet = trk.match_batch_arrays(refs, curs, T_init=guess)
    out[k] = dict(ctx=ctx, trk=trk, refs=refs, curs=curs, quiet=quiet, bad=0, ms=None)


state = {}
for k in range(threads):
    worker(k, state)
go = threading.Barrier(threads)


def run(k):
    s = state[k]
    go.wait()
    t0 = time.perf_counter()
    for _ in range(per_thread):
        o = s["trk"].match_batch_arrays(s["refs"], s["curs"], T_init=guess)
        if not (np.array_equal(o["T"], s["quiet"]["T"]) and np.array_equal(o["information"], s["quiet"]["information"])):
            s["bad"] += 1
    s["ms"] = (time.perf_counter() - t0) / per_thread * 1e3


ts = [threading.Thread(target=run, args=(k,)) for k in range(threads)]
t0 = time.perf_counter()
for t in ts:
    t.start()
for t in ts:
    t.join()
wall = time.perf_counter() - t0
timeouts = sum(state[k]["ctx"].counter("resident_timeouts") for k in range(threads))
launches = sum(state[k]["ctx"].counter("resident_launches") for k in range(threads))
print("%d threads x %d matches of %d pair(s), front-end configuration: %.3f ms per match per thread (%s), %.0f matches/s in total; "
      "resident launches %d, time-outs %d, results differing from the idle device's %d"
      % (threads, per_thread, pairs, np.mean([state[k]["ms"] for k in range(threads)]), ", ".join("%.3f" % state[k]["ms"] for k in range(threads)),
         threads * per_thread / wall, launches, timeouts, sum(state[k]["bad"] for k in range(threads))))
