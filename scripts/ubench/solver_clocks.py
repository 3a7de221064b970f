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
buf = (C.c_ulonglong * 64)()
ctx.set_option("resident", 0)
for _ in range(3):
    out = trk.match_batch_arrays(refs, curs)
L.dvo_hip_debug_solver_clocks(buf, 1)
gbuf = (C.c_ulonglong * 16)()
L.dvo_hip_debug_gn_clocks(gbuf, 1)
t0 = time.perf_counter()
R = 20
for _ in range(R):
    out = trk.match_batch_arrays(refs, curs)
dt = (time.perf_counter() - t0) / R
L.dvo_hip_debug_solver_clocks(buf, 0)
v = np.array(list(buf), dtype=np.float64).reshape(4, 16)
names = ["load state", "stage 3 (reduce partials)", "fused LL", "gn_step (lane 0)", "publish", "store state"]
print("%d pairs: match %.3f ms; iterations %s" % (n, dt * 1e3, out["n_iterations"][:4]))
for row, width in enumerate((640, 320, 160, 80)):
    calls = v[row, 15]
    if calls == 0:
        continue
    print("level %d pixels wide: %d active solver steps of the pairs 0, 64, 128, ... over %d matches; us per step: %s; total %.2f"
          % (width, calls, R, "  ".join("%s %.2f" % (nm, v[row, i] / calls * 0.01) for i, nm in enumerate(names)), v[row, :6].sum() / calls * 0.01))
L.dvo_hip_debug_gn_clocks(gbuf, 0)
g = np.array(list(gbuf), dtype=np.float64)
if g[0] > 0:
    stages = ["record initialised", "precision, log det, prior", "contraction", "6x6 solve", "exp, product, K T", "record, A_last, inverse, product"]
    print("gn_step of pair 0, %d calls, us per call: %s; total %.2f" % (g[0], "  ".join("%s %.2f" % (nm, g[i + 1] / g[0] * 0.01) for i, nm in enumerate(stages)), g[1:7].sum() / g[0] * 0.01))
