#!/usr/bin/env python
"""The reference's own executable (dvo_benchmark/src/benchmark_slam.cpp, compiled unmodified, tests/dropin/Makefile) on a synthetic
TUM-layout folder: once linked against this engine's facade (MI355X), once against the reference's own dvo_core (CPU twin), same
command line.  Prints wall time, the executable's own "match" stopwatch (mean time of KeyframeTracker::update per frame, printed by
the reference code every 100 frames) and the distance between the two trajectory files."""
import os, re, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from dvo_slam_amd import datagen, tum

n = int(sys.argv[1]) if len(sys.argv) > 1 else 201
build = os.path.join(ROOT, "tests", "dropin", "_build")
with tempfile.TemporaryDirectory() as root:
    seq = datagen.synth_sequence(31, n, 640, 480, depth_noise=2.0, grey_noise=4.0, exposure=0.02)
    tum.write_dataset(root, seq["grey"], seq["depth"], seq["poses"])
    out = {}
    for name in ("benchmark_slam", "benchmark_slam_ref"):
        exe = os.path.join(build, name)
        if not os.path.exists(exe):
            print(name, "not built"); continue
        traj = os.path.join(root, name + ".txt")
        t0 = time.perf_counter()
        p = subprocess.run([exe, "_rgbdpair_file:=%s/assoc.txt" % root, "_groundtruth_file:=%s/groundtruth.txt" % root,
                            "_estimate_trajectory:=true", "_trajectory_file:=%s" % traj], cwd=root, capture_output=True, text=True)
        dt = time.perf_counter() - t0
        sw = re.findall(r"match: ([0-9.eE+-]+)", p.stderr + p.stdout)
        stamps, poses = tum.read_trajectory(traj)
        out[name] = poses
        print("%-20s %d frames in %.2f s wall (%.1f ms per frame incl. PNG decoding); the executable's own 'match' stopwatch: %s ms per frame"
              % (name, n, dt, dt / n * 1e3, ", ".join("%.3f" % (float(s) * 1e3) for s in sw) or "-"), flush=True)
    if len(out) == 2:
        from oracle import pyoracle as po
        d = max(np.abs(po.se3_log(np.linalg.inv(a) @ b)).max() for a, b in zip(out["benchmark_slam"], out["benchmark_slam_ref"]))
        print("largest pose distance between the two trajectory files: %.2e" % d)
