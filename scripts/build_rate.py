#!/usr/bin/env python
"""Standalone rate of the frame build (nothing else on the GPU): 128 frames of 640x480 re-ingested from HBM-resident raw planes in
one role, incl. the role planes of levels 1-3, timed around upload-free calls + a device synchronisation."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import dvo_slam_amd as d
from dvo_slam_amd import datagen

W, H, n = 640, 480, 128
b = datagen.synth_batch(0, n, W, H)
dev = torch.device("cuda", 0)
grey = torch.from_numpy(b["grey_ref"]).to(dev)
depth = torch.from_numpy(b["depth_ref"].view(np.int16)).to(dev)
torch.cuda.synchronize()
ctx = d.Context(0)
cam = d.RgbdCameraPyramid(W, H, b["K"], ctx); cam.build(4)
g = d.device_pointer_array([grey[i].data_ptr() for i in range(n)])
z = d.device_pointer_array([depth[i].data_ptr() for i in range(n)])
frames = d.FrameSet([cam.create_raw_device(grey[i].data_ptr(), depth[i].data_ptr()) for i in range(n)])
cfg = d.Config(FirstLevel=3, LastLevel=0)
px = W * H * n
for role, bytes_px in (("current", 40.4), ("reference", 32.7)):
    for _ in range(3):
        d.update_raw_device_batch(frames, g, z, role=role, config=cfg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        d.update_raw_device_batch(frames, g, z, role=role, config=cfg)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print("role %-9s: %.3f ms per %d frames = %.2f us per frame, %.0f GB/s at %.1f B per level-0 pixel" % (role, ms, n, ms * 1e3 / n, bytes_px * px / ms / 1e6, bytes_px))
