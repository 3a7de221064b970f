#!/usr/bin/env python
"""A/B timing of the batched match under library options on the GPU box, with a bit-exact comparison of the results.
usage: ab_match.py <option> <v0,v1,...> [pairs ...]      e.g.  ab_match.py fuse_solver 0,1 1 16 128"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvo_slam_amd as d            # noqa: E402
from dvo_slam_amd import datagen    # noqa: E402

W, H = 640, 480


def main():
    key = sys.argv[1]
    values = [int(v) for v in sys.argv[2].split(",")]
    sizes = [int(a) for a in sys.argv[3:]] or [1, 128]
    nmax = max(sizes)
    ctx = d.default_context()
    b = datagen.synth_batch(0, nmax, W, H)
    cam = d.RgbdCameraPyramid(W, H, b["K"], ctx)
    cam.build(4)
    refs = [cam.create_raw(b["grey_ref"][i], b["depth_ref"][i]) for i in range(nmax)]
    curs = [cam.create_raw(b["grey_cur"][i], b["depth_cur"][i]) for i in range(nmax)]
    trk = d.DenseTracker(d.Config(FirstLevel=3, LastLevel=0), ctx)
    for n in sizes:
        base = None
        for v in values:
            ctx.set_option(key, v)
            out = trk.match_batch_arrays(refs[:n], curs[:n])
            ts = []
            for _ in range(15):
                t0 = time.perf_counter()
                out = trk.match_batch_arrays(refs[:n], curs[:n])
                ts.append((time.perf_counter() - t0) * 1e3)
            raw = b"".join(np.ascontiguousarray(out[k]).tobytes() for k in ("T", "information", "loglik", "n_iterations"))
            if base is None:
                base = raw
            same = "bit-identical" if raw == base else "DIFFERENT"
            print("pairs=%4d %s=%d  median %8.3f ms  min %8.3f ms  (%.0f alignments/s)  results %s vs first" %
                  (n, key, v, np.median(ts), np.min(ts), n / np.median(ts) * 1e3, same), flush=True)


if __name__ == "__main__":
    main()
