#!/usr/bin/env python
"""Round 5: the streaming loop at batch sizes whose first level runs in the resident kernel beside the background ingest (65 ... 112
pairs): steps, resident launches and -- what must stay zero -- groups that timed out and sent their batch to the launch path.
usage: r5_resident_stream.py <steps> <pairs> [<pairs> ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                         # noqa: E402
import dvo_slam_amd as d            # noqa: E402
from dvo_slam_amd import datagen    # noqa: E402
from dvo_slam_amd.stream import StreamPipeline   # noqa: E402

W, H = 640, 480
steps = int(sys.argv[1])
dev = torch.device("cuda", 0)
for B in [int(a) for a in sys.argv[2:]]:
    b = datagen.synth_batch(0, B, W, H, nthreads=min(32, os.cpu_count() or 8))
    grey = torch.from_numpy(np.concatenate([b["grey_ref"], b["grey_cur"]])).to(dev)
    depth = torch.from_numpy(np.concatenate([b["depth_ref"], b["depth_cur"]]).view(np.int16)).to(dev)
    torch.cuda.synchronize()
    gp = [grey[i].data_ptr() for i in range(2 * B)]
    zp = [depth[i].data_ptr() for i in range(2 * B)]
    ctx = d.Context(0)
    ctx.set_option("build_workgroups", 256)
    cam = d.RgbdCameraPyramid(W, H, b["K"], ctx)
    cam.build(4)
    sets = [[cam.create_raw_device(gp[i], zp[i]) for i in range(2 * B)] for _ in range(2)]
    cfg = d.Config(FirstLevel=3, LastLevel=0, MaxIterationsPerLevel=100, Precision=5e-7, Mu=0.0)
    pipe = StreamPipeline(ctx, cfg, [d.FrameSet(fs[:B]) for fs in sets], [d.FrameSet(fs[B:]) for fs in sets], gp[:B], zp[:B], gp[B:], zp[B:])
    pipe.step(now=None, nxt=0)
    for j in range(3):
        pipe.step(now=j % 2, nxt=(j + 1) % 2)
    torch.cuda.synchronize()
    k0 = ctx.counter("resident_launches"), ctx.counter("resident_timeouts")
    t0 = time.perf_counter()
    worst = 0.0
    for j in range(3, 3 + steps):
        t1 = time.perf_counter()
        res = pipe.step(now=j % 2, nxt=(j + 1) % 2)
        worst = max(worst, time.perf_counter() - t1)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    T = res["transformation"].reshape(B, 4, 4)
    print("%4d pairs: %d steps, %.3f ms per step (slowest %.3f ms); resident launches %d, timed out %d; results finite: %s" % (
        B, steps, ms, worst * 1e3, ctx.counter("resident_launches") - k0[0], ctx.counter("resident_timeouts") - k0[1], bool(np.isfinite(T).all())))
    del pipe, sets, ctx
