#!/usr/bin/env python
"""A/B of the resident match kernel against the launch-per-iteration path: same pairs, both paths, results and time per match.
usage: resident_ab.py [pairs ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dvo_slam_amd as d
from dvo_slam_amd import datagen
from dvo_slam_amd.parallel import twists_of

sizes = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [1, 2, 4, 16, 128]
ctx = d.default_context()
if "--cooperative" in sys.argv:
    ctx.set_option("resident_cooperative", 1)      # groups through hipLaunchCooperativeKernel
nmax = max(sizes)
seed0 = int([a.split("=")[1] for a in sys.argv if a.startswith("--seed=")][0]) if any(a.startswith("--seed=") for a in sys.argv) else 0
b = datagen.synth_batch(seed0, nmax, 640, 480)
cam = d.RgbdCameraPyramid(640, 480, b["K"], ctx); cam.build(4)
refs = [cam.create_raw(b["grey_ref"][i], b["depth_ref"][i]) for i in range(nmax)]
curs = [cam.create_raw(b["grey_cur"][i], b["depth_cur"][i]) for i in range(nmax)]
trk = d.DenseTracker(d.Config(FirstLevel=3, LastLevel=0, MaxIterationsPerLevel=100, Precision=5e-7, Mu=0.0), ctx)


def run(n, reps):
    for _ in range(3):
        out = trk.match_batch_arrays(refs[:n], curs[:n])
    t0 = time.perf_counter()
    for _ in range(reps):
        out = trk.match_batch_arrays(refs[:n], curs[:n])
    return out, (time.perf_counter() - t0) / reps * 1e3


for n in sizes:
    reps = 50 if n <= 16 else 10
    ctx.set_option("resident", 0)
    base, t_base = run(n, reps)
    line = "pairs %4d  launches %.3f ms" % (n, t_base)
    for mode, rows in ((-1, 0), (1, 0)):
        ctx.set_option("resident", mode)
        ctx.set_option("resident_rows", rows)
        out, t = run(n, reps)
        dT = np.abs(twists_of(np.linalg.inv(out["T"]) @ base["T"])).max()
        dit = np.abs(out["n_iterations"].astype(int) - base["n_iterations"].astype(int)).max()
        line += "   resident=%2d %.3f ms (|dtwist| %.1e, |diters| %d)" % (mode, t, dT, dit)
    print(line, flush=True)
    if "--host" in sys.argv:
        for mode in (0, -1):
            ctx.set_option("resident", mode)
            keys = ("host_batches", "host_ns_prepare", "host_ns_enqueue", "host_ns_wait", "host_ns_finish")
            c0 = [ctx.counter(k) for k in keys]
            _, t = run(n, reps)
            c1 = [ctx.counter(k) for k in keys]
            nb = c1[0] - c0[0]
            print("      resident=%2d: %.3f ms per call; inside the library, us per batch: prepare %.1f, enqueue %.1f, wait %.1f, finish %.1f"
                  % ((mode, t) + tuple((b - a) / nb / 1e3 for a, b in zip(c0[1:], c1[1:]))), flush=True)
