#!/usr/bin/env python
"""Where a solver step's time goes (experiment build with -DDVO_SOLVER_CLOCKS, see solver_kernels.hip): one 640x480 pair,
BASELINE config 2.  Run with DVO_HIP_LIBRARY=scripts/ubench/_build/libdvo_hip_clk.so."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import dvo_slam_amd as d
from dvo_slam_amd import datagen, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ctx = d.default_context()
b = datagen.synth_batch(0, n, 640, 480)
cam = d.RgbdCameraPyramid(640, 480, b["K"], ctx); cam.build(4)
refs = [cam.create_raw(b["grey_ref"][i], b["depth_ref"][i]) for i in range(n)]
curs = [cam.create_raw(b["grey_cur"][i], b["depth_cur"][i]) for i in range(n)]
trk = d.DenseTracker(d.Config(FirstLevel=3, LastLevel=0, MaxIterationsPerLevel=100, Precision=5e-7, Mu=0.0), ctx)
L = C.CDLL(_lib.LIB_PATH)
buf = (C.c_ulonglong * 16)()
for _ in range(3):
    out = trk.match_batch_arrays(refs, curs)
L.dvo_hip_debug_solver_clocks(buf, 1)
t0 = time.perf_counter()
R = 20
for _ in range(R):
    out = trk.match_batch_arrays(refs, curs)
dt = (time.perf_counter() - t0) / R
L.dvo_hip_debug_solver_clocks(buf, 0)
v = np.array(list(buf), dtype=np.float64)
calls = v[15]
names = ["load state", "stage 3 (reduce partials)", "fused LL", "gn_step (lane 0)", "publish", "store state"]
print("match %.3f ms; %d active solver steps of pair 0 over %d matches (%.1f per match), iterations %s" % (dt * 1e3, calls, R, calls / R, out["n_iterations"][:4]))
for i, nm in enumerate(names):
    print("  %-28s %7.2f us per step" % (nm, v[i] / calls * 0.01))
print("  total %.2f us per step" % (v[:6].sum() / calls * 0.01))
