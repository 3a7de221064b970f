#!/usr/bin/env python
"""Tuning sweep on the GPU box: duration of the fused residual/Jacobian/reduce kernel vs tile height and batch size,
and batched match time vs host polling cadence.  Prints a table; results feed DESIGN.md."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvo_slam_amd as d            # noqa: E402
from dvo_slam_amd import datagen    # noqa: E402

W, H = 640, 480


def main():
    nmax = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    ctx = d.default_context()
    b = datagen.synth_batch(0, nmax, W, H)
    cam = d.RgbdCameraPyramid(W, H, b["K"], ctx)
    cam.build(4)
    refs = [cam.create_raw(b["grey_ref"][i], b["depth_ref"][i]) for i in range(nmax)]
    curs = [cam.create_raw(b["grey_cur"][i], b["depth_cur"][i]) for i in range(nmax)]
    trk = d.DenseTracker(d.Config(FirstLevel=3, LastLevel=0), ctx)
    rows = []
    for variant in (5,):
        ctx.set_option("variant", variant)
        for n in [x for x in (1, 128) if x <= nmax]:
            for level in (0, 1):
                for rpw in (4, 8, 16):
                    ctx.set_option("rows_per_wave", rpw)
                    ms = trk.time_residual_kernel(refs[:n], curs[:n], level, reps=10)
                    px = (W >> level) * (H >> level) * n
                    rows.append(dict(kind="variant", variant=variant, pairs=n, level=level, rows_per_wave=rpw, ms=ms, gbps=40.0 * px / (ms * 1e-3) / 1e9))
                    print("variant=%d pairs=%4d level=%d rpw=%2d  %9.4f ms  %8.1f GB/s (40 B/px)" % (variant, n, level, rpw, ms, rows[-1]["gbps"]), flush=True)
    ctx.set_option("variant", int(os.environ.get("DVO_VARIANT", "5")))
    for n in [x for x in (1, 8, 32, 128, 256) if x <= nmax and os.environ.get("FULL_SWEEP", "0") == "1"]:
        for level in (0, 1, 2, 3):
            for rpw in (1, 2, 4, 8, 16):
                ctx.set_option("rows_per_wave", rpw)
                ms = trk.time_residual_kernel(refs[:n], curs[:n], level, reps=10)
                px = (W >> level) * (H >> level) * n
                rows.append(dict(kind="kernel", pairs=n, level=level, rows_per_wave=rpw, ms=ms, gbps=40.0 * px / (ms * 1e-3) / 1e9))
                print("kernel pairs=%4d level=%d rpw=%2d  %9.4f ms  %8.1f GB/s (40 B/px)" % (n, level, rpw, ms, rows[-1]["gbps"]), flush=True)
    ctx.set_option("rows_per_wave", 0)
    for n in [x for x in (1, 128) if x <= nmax]:
        res = [d.Result() for _ in range(n)]
        for ips in (1, 2, 4, 8, 16):
            ctx.set_option("iters_per_sync", ips)
            trk.match_batch(refs[:n], curs[:n], res)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                trk.match_batch(refs[:n], curs[:n], res)
                ts.append((time.perf_counter() - t0) * 1e3)
            rows.append(dict(kind="match", pairs=n, iters_per_sync=ips, ms=float(np.median(ts))))
            print("match  pairs=%4d iters_per_sync=%2d  %9.3f ms  (%.1f alignments/s)" % (n, ips, rows[-1]["ms"], n / rows[-1]["ms"] * 1e3), flush=True)
    ctx.set_option("iters_per_sync", 0)
    # iterations actually taken
    res = [d.Result() for _ in range(min(nmax, 16))]
    trk.match_batch(refs[:len(res)], curs[:len(res)], res, with_stats=True)
    its = np.array([[len(L.Iterations) for L in r.Statistics.Levels] for r in res])
    print("iterations per level (rows = pairs, cols = levels 3..0):\n", its, "\nmean", its.mean(0))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(rows, open("gpurun_out/sweep.json", "w"))


if __name__ == "__main__":
    main()
