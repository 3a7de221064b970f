#!/usr/bin/env python
"""BASELINE config 5 on the GPU box: 1280x960, 5 pyramid levels (FirstLevel=4, LastLevel=0).  Sweep-kernel time per level and
tile height, and whole-match time for 1 / 8 / 32 pairs."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvo_slam_amd as d            # noqa: E402
from dvo_slam_amd import datagen    # noqa: E402

W, H = 1280, 960


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    ctx = d.default_context()
    b = datagen.synth_batch(4321, n, W, H)
    cam = d.RgbdCameraPyramid(W, H, b["K"], ctx)
    cam.build(5)
    refs = [cam.create_raw(b["grey_ref"][i], b["depth_ref"][i]) for i in range(n)]
    curs = [cam.create_raw(b["grey_cur"][i], b["depth_cur"][i]) for i in range(n)]
    trk = d.DenseTracker(d.Config(FirstLevel=4, LastLevel=0), ctx)
    print("sweep kernel (k_residual_reduce_mfma), %d pairs per launch: ms per launch (GB/s at 40 B/px)" % n)
    for level in range(5):
        row = []
        for rpw in (1, 2, 4, 8, 16):
            ctx.set_option("rows_per_wave", rpw)
            ms = trk.time_residual_kernel(refs, curs, level, reps=10)
            px = (W >> level) * (H >> level) * n
            row.append("rpw%2d %.4f (%5.0f)" % (rpw, ms, 40.0 * px / (ms * 1e-3) / 1e9))
        print("  level %d (%4dx%3d): %s" % (level, W >> level, H >> level, "  ".join(row)), flush=True)
    ctx.set_option("rows_per_wave", 0)
    for m in (1, 8, n):
        out = trk.match_batch_arrays(refs[:m], curs[:m])
        ts = []
        for _ in range(8):
            t0 = time.perf_counter()
            out = trk.match_batch_arrays(refs[:m], curs[:m])
            ts.append((time.perf_counter() - t0) * 1e3)
        print("match %3d pairs: median %.3f ms (%.0f alignments/s), iterations %s" % (m, np.median(ts), m / np.median(ts) * 1e3, out["n_iterations"][:4]), flush=True)


if __name__ == "__main__":
    main()
