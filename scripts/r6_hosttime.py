#!/usr/bin/env python
"""Where the HOST thread's time of a one-lane streaming step goes (1024 pairs): the two ingest calls (asynchronous: host time only), the
match (host_ns_* counters of the library: before the first launch / enqueueing / waiting for the device / afterwards)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import dvo_slam_amd as d
from dvo_slam_amd import datagen

W, H, B = 640, 480, int(sys.argv[1]) if len(sys.argv) > 1 else 1024
b = datagen.synth_batch(0, B, W, H, nthreads=32)
dev = torch.device("cuda", 0)
grey = torch.from_numpy(np.concatenate([b["grey_ref"], b["grey_cur"]])).to(dev)
depth = torch.from_numpy(np.concatenate([b["depth_ref"], b["depth_cur"]]).view(np.int16)).to(dev)
torch.cuda.synchronize()
ctx = d.Context(0)
ctx.set_option("build_workgroups", 256)
cam = d.RgbdCameraPyramid(W, H, b["K"], ctx); cam.build(4)
gp = [grey[i].data_ptr() for i in range(2 * B)]; zp = [depth[i].data_ptr() for i in range(2 * B)]
sets = [[cam.create_raw_device(gp[i], zp[i]) for i in range(2 * B)] for _ in range(2)]
rs = [d.FrameSet(s[:B]) for s in sets]; cs = [d.FrameSet(s[B:]) for s in sets]
g_ref, z_ref = d.device_pointer_array(gp[:B]), d.device_pointer_array(zp[:B])
g_cur, z_cur = d.device_pointer_array(gp[B:]), d.device_pointer_array(zp[B:])
cfg = d.Config(FirstLevel=3, LastLevel=0)
trk = d.DenseTracker(cfg, ctx)
d.update_raw_device_batch(rs[0], g_ref, z_ref, role="reference", config=cfg)
d.update_raw_device_batch(cs[0], g_cur, z_cur, role="current", config=cfg)
keys = ("host_ns_prepare", "host_ns_enqueue", "host_ns_wait", "host_ns_finish")
t_ing = [0.0, 0.0]; t_match = 0.0; n = 12
for k in range(n + 2):
    if k == 2:
        base = {key: ctx.counter(key) for key in keys}; t_ing = [0.0, 0.0]; t_match = 0.0
        torch.cuda.synchronize(); t_all = time.perf_counter()
    nxt = (k + 1) % 2
    t0 = time.perf_counter()
    d.update_raw_device_batch(rs[nxt], g_ref, z_ref, role="reference", config=cfg)
    t1 = time.perf_counter()
    d.update_raw_device_batch(cs[nxt], g_cur, z_cur, role="current", config=cfg)
    t2 = time.perf_counter()
    trk.match_batch_arrays(rs[k % 2], cs[k % 2])
    t3 = time.perf_counter()
    t_ing[0] += t1 - t0; t_ing[1] += t2 - t1; t_match += t3 - t2
torch.cuda.synchronize(); t_all = time.perf_counter() - t_all
print("%d pairs: step %.3f ms; host time per step: ingest call (reference frames) %.3f ms, (current frames) %.3f ms, match call %.3f ms" % (B, t_all / n * 1e3, t_ing[0] / n * 1e3, t_ing[1] / n * 1e3, t_match / n * 1e3))
print("   inside the match call: " + ", ".join("%s %.3f ms" % (key[8:], (ctx.counter(key) - base[key]) / n * 1e-6) for key in keys))
