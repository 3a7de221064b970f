#!/usr/bin/env python
"""Where an iteration of the resident match kernel spends its time (experiment build with -DDVO_RESIDENT_CLOCKS; run with
DVO_HIP_LIBRARY=scripts/ubench/_build/libdvo_hip_clk.so).  usage: resident_clocks.py [pairs [group [last_level]]]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import dvo_slam_amd as d
from dvo_slam_amd import datagen, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
group = int(sys.argv[2]) if len(sys.argv) > 2 else 0
last = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ctx = d.default_context()
ctx.set_option("resident", 1)
ctx.set_option("resident_group", group)
b = datagen.synth_batch(0, n, 640, 480)
cam = d.RgbdCameraPyramid(640, 480, b["K"], ctx); cam.build(4)
refs = [cam.create_raw(b["grey_ref"][i], b["depth_ref"][i]) for i in range(n)]
curs = [cam.create_raw(b["grey_cur"][i], b["depth_cur"][i]) for i in range(n)]
trk = d.DenseTracker(d.Config(FirstLevel=3, LastLevel=last, MaxIterationsPerLevel=100, Precision=5e-7, Mu=0.0), ctx)
L = C.CDLL(_lib.LIB_PATH)
buf = (C.c_ulonglong * 32)()
for _ in range(3):
    out = trk.match_batch_arrays(refs, curs)
L.dvo_hip_debug_resident_clocks(buf, 1)
R = 20
t0 = time.perf_counter()
for _ in range(R):
    out = trk.match_batch_arrays(refs, curs)
dt = (time.perf_counter() - t0) / R
L.dvo_hip_debug_resident_clocks(buf, 0)
v = np.array(list(buf), dtype=np.float64) * 0.01 / R
names = ["level begin / prologue", "sweep", "wait + fold", "exchange", "log-likelihood", "-", "solver", "epilogue"]
print("pairs %d group %d: match %.3f ms host-side; %.1f iterations per match; kernel sections of workgroup 0, us per match:" % (n, group, dt * 1e3, buf[15] / R))
for i, nm in enumerate(names):
    print("  %-24s %8.1f" % (nm, v[i]))
print("  %-24s %8.1f" % ("sum", v[:8].sum()))
print("inside gn_step (us per match):")
for i, nm in enumerate(["record initialised", "precision, log det, prior", "contraction", "6x6 solve", "record / A_last copies", "exp, inverse, products, K T"]):
    print("  %-28s %8.1f" % (nm, v[17 + i]))
