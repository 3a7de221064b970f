#!/usr/bin/env python
"""Experiment: the 128-pair step of bench.py split over T host threads, one context each (the library's threading model:
one context per host thread), so that one thread's latency-bound coarse levels overlap the other's bandwidth-bound fine levels."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import dvo_slam_amd as d
from dvo_slam_amd import datagen

W, H, B = 640, 480, int(os.environ.get("DVO_PAIRS", "128"))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
b = datagen.synth_batch(0, B, W, H)
dev = torch.device("cuda", 0)
grey = torch.from_numpy(np.concatenate([b["grey_ref"], b["grey_cur"]])).to(dev)
depth = torch.from_numpy(np.concatenate([b["depth_ref"], b["depth_cur"]]).view(np.int16)).to(dev)
torch.cuda.synchronize()
cfg = d.Config(FirstLevel=3, LastLevel=0, MaxIterationsPerLevel=100, Precision=5e-7, Mu=0.0)


class Worker:
    def __init__(self, idx):   # idx: pair indices of this worker
        self.ctx = d.Context(0)
        self.cam = d.RgbdCameraPyramid(W, H, b["K"], self.ctx)
        self.cam.build(4)
        self.n = len(idx)
        self.gr = [grey[i].data_ptr() for i in idx]
        self.zr = [depth[i].data_ptr() for i in idx]
        self.gc = [grey[B + i].data_ptr() for i in idx]
        self.zc = [depth[B + i].data_ptr() for i in idx]
        self.sets = [([self.cam.create_raw_device(g, z) for g, z in zip(self.gr, self.zr)],
                      [self.cam.create_raw_device(g, z) for g, z in zip(self.gc, self.zc)]) for _ in range(2)]
        self.trk = d.DenseTracker(cfg, self.ctx)
        self.k = 0
        self.build(0)

    def build(self, k):
        r, c = self.sets[k]
        d.update_raw_device_batch(r, self.gr, self.zr, role="reference", config=cfg)
        d.update_raw_device_batch(c, self.gc, self.zc, role="current", config=cfg)

    def step(self):
        k = self.k % 2
        self.k += 1
        self.build((k + 1) % 2)
        r, c = self.sets[k]
        self.out = self.trk.match_batch_arrays(r, c)


for T in [int(x) for x in sys.argv[1].split(",")]:
    workers = [Worker(list(range(t, B, T))) for t in range(T)]
    go = threading.Barrier(T + 1)
    def run(w, n):
        go.wait()
        for _ in range(n):
            w.step()
    for n in (2, steps):   # warm-up, then timed
        ths = [threading.Thread(target=run, args=(w, n)) for w in workers]
        for t in ths:
            t.start()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        go.wait()
        for t in ths:
            t.join()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    print("threads %d: %.3f ms per %d-pair step, %.0f alignments/s" % (T, el / steps * 1e3, B, B * steps / el))
    del workers
