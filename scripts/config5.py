#!/usr/bin/env python
"""BASELINE config 5 on the GPU box: 1280x960, 5 pyramid levels (FirstLevel=4, LastLevel=0) -- the "LDS-tile occupancy sweep": every
schedule of the level sweep with its tile shape, LDS per workgroup and wavefronts per SIMD (static properties of the build, from
hipcc -Rpass-analysis=kernel-resource-usage), timed per level at the converged transform, and whole-match times for 1 / 8 / 32 pairs.
  variant 5: gathered taps, f32 Gram on the matrix cores      tile 64 x (4 rows_per_wave), 16.9 KB LDS, 64 VGPRs -> 7 wavefronts / SIMD
  variant 6: {I,Z} window staged in LDS, f32 Gram              tile 64 x 16, window 80 x 30 x 8 B = 19.2 KB + 16.9 KB, 84 VGPRs -> 4 / SIMD
  variant 7: window + f16 hi/lo Gram on the matrix pipe        tile 64 x 16, 19.2 KB + 10.2 KB half-row slabs = 29.5 KB, 88-94 VGPRs -> 5 / SIMD (round 3's default)
  variant 8: window (84 x 28 at a pitch of 96) + contracted arithmetic + f16 Gram, operands hi / lo in turns   tile 64 x 16, 21.5 KB + 10.2 KB = 31.8 KB, 89 VGPRs -> 5 / SIMD (default)
  variant 9: the same with the operands moved by v_permlane32_swap                                             tile 64 x 16, 31.8 KB, 96 VGPRs -> 5 / SIMD
(levels narrower than 64 pixels x k walk the level as one row of pixels with variant 5 whatever the option says)"""
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
    print(__doc__.split("\n", 1)[1])
    print("level sweep, %d pairs per launch, converged transform, weights on: ms per launch (GB/s at the 40 algorithmic B/px)" % n)
    schedules = [("v5 64x8", 5, 2), ("v5 64x16", 5, 4), ("v5 64x32", 5, 8), ("v5 64x64", 5, 16), ("v6 64x16", 6, 0), ("v7 64x16", 7, 0), ("v8 64x16", 8, 0), ("v9 64x16", 9, 0)]
    for level in range(5):
        row = []
        for label, variant, rpw in schedules:
            ctx.set_option("variant", variant)
            ctx.set_option("rows_per_wave", rpw)
            ms = min(trk.time_residual_kernel(refs, curs, level, reps=10, warm_iterations=3) for _ in range(2))
            px = (W >> level) * (H >> level) * n
            row.append("%s %.4f (%4.0f)" % (label, ms, 40.0 * px / (ms * 1e-3) / 1e9))
        print("  level %d (%4dx%3d): %s" % (level, W >> level, H >> level, "  ".join(row)), flush=True)
    ctx.set_option("rows_per_wave", 0)
    for variant in (5, 7, 8):
        ctx.set_option("variant", variant)
        for m in (1, 8, n):
            out = trk.match_batch_arrays(refs[:m], curs[:m])
            ts = []
            for _ in range(8):
                t0 = time.perf_counter()
                out = trk.match_batch_arrays(refs[:m], curs[:m])
                ts.append((time.perf_counter() - t0) * 1e3)
            print("variant %d, match %3d pairs: median %.3f ms (%.0f alignments/s), iterations %s, window fall-back lanes so far %d"
                  % (variant, m, np.median(ts), m / np.median(ts) * 1e3, out["n_iterations"][:4], ctx.counter("window_fallbacks")), flush=True)
    ctx.set_option("variant", 8)


if __name__ == "__main__":
    main()
