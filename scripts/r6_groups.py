#!/usr/bin/env python
"""Experiment (round 6): bench.py's streaming loop of B pairs per step split over G host threads with a context each on ONE GPU -- group g
takes the pairs g, g + G, ... and runs its own dvo_stream_step (the tuned loop: role-aware ingest of the next batch in the background,
capped build grid).  One group's latency-bound solver steps and coarse sweeps run beside the other's sweeps.
usage: r6_groups.py <pairs> <groups,groups,...> [steps] [build cap per context, comma list aligned with groups]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import dvo_slam_amd as d
from dvo_slam_amd import datagen
from dvo_slam_amd.stream import StreamPipeline

W, H = 640, 480
B = int(sys.argv[1])
groups = [int(x) for x in sys.argv[2].split(",")]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
caps = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [256] * len(groups)
SYNC = len(sys.argv) > 5 and sys.argv[5] == "sync"      # the groups meet after every step (what a synchronous batch call does) instead of running free
b = datagen.synth_batch(0, B, W, H, nthreads=min(32, os.cpu_count() or 8))
dev = torch.device("cuda", 0)
grey = torch.from_numpy(np.concatenate([b["grey_ref"], b["grey_cur"]])).to(dev)
depth = torch.from_numpy(np.concatenate([b["depth_ref"], b["depth_cur"]]).view(np.int16)).to(dev)
torch.cuda.synchronize()
cfg = d.Config(FirstLevel=3, LastLevel=0, MaxIterationsPerLevel=100, Precision=5e-7, Mu=0.0)


class Worker:
    def __init__(self, idx, cap):
        self.ctx = d.Context(0)
        self.ctx.set_option("build_workgroups", cap)
        self.cam = d.RgbdCameraPyramid(W, H, b["K"], self.ctx)
        self.cam.build(4)
        n = len(idx)
        gr, zr = [grey[i].data_ptr() for i in idx], [depth[i].data_ptr() for i in idx]
        gc, zc = [grey[B + i].data_ptr() for i in idx], [depth[B + i].data_ptr() for i in idx]
        sets = [[self.cam.create_raw_device(g, z) for g, z in zip(gr + gc, zr + zc)] for _ in range(2)]
        self.pipe = StreamPipeline(self.ctx, cfg, [d.FrameSet(fs[:n]) for fs in sets], [d.FrameSet(fs[n:]) for fs in sets], gr, zr, gc, zc)
        self.k = 0
        self.pipe.step(now=None, nxt=0)

    def step(self):
        k = self.k % 2
        self.k += 1
        self.out = self.pipe.step(now=k, nxt=(k + 1) % 2)


for G, cap in zip(groups, caps):
    workers = [Worker(list(range(g, B, G)), cap) for g in range(G)]
    go = threading.Barrier(G + 1)

    every = threading.Barrier(G)

    def run(w, n):
        go.wait()
        for _ in range(n):
            w.step()
            if SYNC:
                every.wait()
    for n in (3, steps):   # warm-up, then timed
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
    print("%d pairs per step, %d group(s)%s, build cap %d each: %.3f ms per step, %.0f alignments/s" % (B, G, " meeting after every step" if SYNC else "", cap, el / steps * 1e3, B * steps / el), flush=True)
    del workers
