#!/usr/bin/env python
"""Round-2 A/B on the GPU box: the sweep with the log-likelihood in its own launch (residual pairs in registers, hand-off of the
precision) against the round-1 schedule (residual pairs stored, second sweep).  Kernel time at the converged transform with the
weights on, the stream yardsticks, whole-match time for 1 / 16 / 128 pairs, and the distance between the two paths' results."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvo_slam_amd as d            # noqa: E402
from dvo_slam_amd import datagen    # noqa: E402

W, H = 640, 480
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ctx = d.default_context()
b = datagen.synth_batch(0, N, W, H)
cam = d.RgbdCameraPyramid(W, H, b["K"], ctx)
cam.build(4)
refs = [cam.create_raw(b["grey_ref"][i], b["depth_ref"][i]) for i in range(N)]
curs = [cam.create_raw(b["grey_cur"][i], b["depth_cur"][i]) for i in range(N)]
trk = d.DenseTracker(d.Config(FirstLevel=3, LastLevel=0), ctx)

print("stream yardstick, %d pairs, level 0: read-only %.4f ms, read+write %.4f ms" % (
    N, trk.time_stream_mix(refs, curs, 0, reps=20), trk.time_stream_mix(refs, curs, 0, reps=20, with_write=True)), flush=True)
for inkernel in (1, 0):
    ctx.set_option("inkernel_ll", inkernel)
    for rpw in (8, 4):
        ctx.set_option("rows_per_wave", rpw)
        for warm in (0, 3):
            ms = [trk.time_residual_kernel(refs, curs, 0, reps=20, warm_iterations=warm) for _ in range(3)]
            print("inkernel_ll=%d rows_per_wave=%d warm=%d: sweep %.4f ms (min of 3; %s) = %.0f GB/s algorithmic = %.3f of 8 TB/s" % (
                inkernel, rpw, warm, min(ms), " ".join("%.4f" % m for m in ms), 40.0 * W * H * N / (min(ms) * 1e-3) / 1e9,
                40.0 * W * H * N / (min(ms) * 1e-3) / 8e12), flush=True)
    ctx.set_option("rows_per_wave", 0)
    for lvl in (1, 2, 3):
        print("   level %d sweep: %.4f ms" % (lvl, trk.time_residual_kernel(refs, curs, lvl, reps=20, warm_iterations=3)), flush=True)

res = {}
for n in (1, 16, N):
    for inkernel in (1, 0):
        ctx.set_option("inkernel_ll", inkernel)
        out = trk.match_batch_arrays(refs[:n], curs[:n])
        ts = []
        for _ in range(15):
            t0 = time.perf_counter()
            out = trk.match_batch_arrays(refs[:n], curs[:n])
            ts.append((time.perf_counter() - t0) * 1e3)
        res[(n, inkernel)] = out
        print("pairs=%4d inkernel_ll=%d  match median %8.3f ms  min %8.3f ms  (%.0f alignments/s)  iterations %.1f per pair" % (
            n, inkernel, np.median(ts), np.min(ts), n / np.median(ts) * 1e3, float(np.mean(out["n_iterations"]))), flush=True)
    a, c = res[(n, 1)], res[(n, 0)]
    dT = max(np.abs(np.linalg.inv(a["T"][i]) @ c["T"][i] - np.eye(4)).max() for i in range(n))
    print("pairs=%4d  max |T_inkernel^-1 T_second - I| = %.3e   iteration counts equal: %s   loglik max rel diff %.3e" % (
        n, dT, bool(np.array_equal(a["n_iterations"], c["n_iterations"])),
        float(np.max(np.abs(a["loglik"] - c["loglik"]) / np.abs(c["loglik"])))), flush=True)
    print("pairs=%4d  nan results: %d" % (n, int(np.isnan(a["T"]).any(axis=(1, 2)).sum())), flush=True)
