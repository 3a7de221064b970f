#!/usr/bin/env python
"""Cost of one frame's life in a tracking loop: create (upload + pyramid) + build(4) + destroy, 640x480."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dvo_slam_amd as d
from dvo_slam_amd import datagen
from oracle import pyoracle as po
ctx = d.default_context()
b = datagen.synth_batch(0, 2, 640, 480)
cam = d.RgbdCameraPyramid(640, 480, b["K"], ctx); cam.build(4)
I = b["grey_ref"][0].astype(np.float32); Z = po.convert_raw_depth(b["depth_ref"][0])
for label, make in (("create (float planes)", lambda: cam.create(I, Z)), ("create_raw (u8 + u16)", lambda: cam.create_raw(b["grey_ref"][0], b["depth_ref"][0]))):
    frames = [make() for _ in range(3)]
    del frames
    t0 = time.perf_counter()
    for _ in range(50):
        f = make()
        f.build(4)
        del f
    print("%-24s %.3f ms per frame (create + build(4) + destroy)" % (label, (time.perf_counter() - t0) / 50 * 1e3))
    keep = []
    t0 = time.perf_counter()
    for _ in range(50):
        f = make(); f.build(4); keep.append(f)
    print("%-24s %.3f ms per frame (create + build(4), frames kept)" % (label, (time.perf_counter() - t0) / 50 * 1e3))
    del keep
