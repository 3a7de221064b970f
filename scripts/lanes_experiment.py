#!/usr/bin/env python
"""How much of a 1024-pair match is serial latency that independent sub-batches on streams of their own would hide?  T host threads,
one context (= one stream) each, align 1024 / T pairs of BASELINE config 4 at the same time; frames are built beforehand.
usage: lanes_experiment.py [total pairs] [rounds]"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvo_slam_amd as d            # noqa: E402
from dvo_slam_amd import datagen    # noqa: E402

total = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
b = datagen.synth_batch(0, 64, 640, 480)
cfg = d.Config(FirstLevel=3, LastLevel=0)
for threads in (1, 2, 4):
    n = total // threads
    state = []
    for k in range(threads):
        ctx = d.Context(0)
        cam = d.RgbdCameraPyramid(640, 480, b["K"], ctx)
        cam.build(4)
        refs = [cam.create_raw(b["grey_ref"][(k * n + i) % 64], b["depth_ref"][(k * n + i) % 64]) for i in range(n)]
        curs = [cam.create_raw(b["grey_cur"][(k * n + i) % 64], b["depth_cur"][(k * n + i) % 64]) for i in range(n)]
        trk = d.DenseTracker(cfg, ctx)
        trk.match_batch_arrays(refs, curs)
        state.append((ctx, trk, refs, curs))
    go = threading.Barrier(threads + 1)
    done = threading.Barrier(threads + 1)

    def run(k):
        ctx, trk, refs, curs = state[k]
        for _ in range(rounds):
            go.wait()
            trk.match_batch_arrays(refs, curs)
            done.wait()
    ts = [threading.Thread(target=run, args=(k,)) for k in range(threads)]
    for t in ts:
        t.start()
    walls = []
    for _ in range(rounds):
        go.wait()
        t0 = time.perf_counter()
        done.wait()
        walls.append((time.perf_counter() - t0) * 1e3)
    for t in ts:
        t.join()
    print("%d stream(s) x %d pairs: %.2f ms per %d pairs (median of %d; min %.2f)" % (threads, n, np.median(walls), total, rounds, min(walls)), flush=True)
    del state
