#!/usr/bin/env python
"""Per-pass time of the CPU side at the finest level of one synthetic 640x480 pair ("single-thread ms/pair and
per-pass breakdown at S640"): the oracle's REF_SSE passes and, when oracle/_ref is built, the reference's own functions on the
same arrays (tests/test_oracle_ref.py pins the two bit for bit).  Runs on the host only."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import common as cm                    # noqa: E402
from oracle import pyoracle as po      # noqa: E402
import test_oracle_ref as tr           # noqa: E402  (array builders of the pin tests)

REPS = 20


def best(f):
    ts = []
    for _ in range(REPS):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3


def main():
    pair = cm.synth(1234, 640, 480)
    oref, ocur = cm.oracle_pyramids(pair, 1)
    _, pts, K, w, h = tr.level_arrays(oref, 0)
    accel, _, _, _, _ = tr.level_arrays(ocur, 0)
    n = len(pts)
    T34 = np.ascontiguousarray(po.se3_exp(-0.5 * pair["xi_true"])[:3], np.float32).reshape(-1)
    fp = tr.fp
    out_p, out_r = np.zeros((n + 2, 12), np.float32), np.zeros((n + 2, 2), np.float32)
    L, R = po.lib(), po.ref_lib()
    wr, wc = np.zeros(8, np.float32), np.zeros(8, np.float32)
    L.oracle_pass_weight_vectors(fp(K), fp(wr), fp(wc))
    m = L.oracle_pass_residuals(po.REF_SSE, n, fp(pts), fp(accel), w, h, fp(K), fp(T34), fp(out_p), fp(out_r))
    res = np.ascontiguousarray(out_r[:m])
    P = tr.f32(2500.0, -300.0, -300.0, 9000.0)
    zero = tr.f32(0, 0)
    wts, S = np.zeros(m, np.float32), np.zeros(4, np.float32)
    J = np.ascontiguousarray(np.random.default_rng(0).normal(size=(m, 12)), np.float32)
    A32, A64 = np.zeros(36, np.float32), np.zeros(36, np.float64)
    rows = [
        ("warp + bilinear sample + residual (pass 1)", lambda: L.oracle_pass_residuals(po.REF_SSE, n, fp(pts), fp(accel), w, h, fp(K), fp(T34), fp(out_p), fp(out_r)),
         R and (lambda: R.ref_compute_residuals(1, n, fp(pts), fp(accel), w, h, fp(K), fp(T34), fp(wr), fp(wc), fp(out_p), fp(out_r)))),
        ("t-distribution weights (pass 2)", lambda: L.oracle_pass_weights(po.REF_SSE, m, fp(res), fp(P), fp(wts)),
         R and (lambda: R.ref_compute_weights(1, m, fp(res), fp(zero), fp(P), fp(wts)))),
        ("2x2 scale (pass 3)", lambda: L.oracle_pass_scale(po.REF_SSE, m, fp(res), fp(wts), fp(S)),
         R and (lambda: R.ref_compute_scale(1, m, fp(res), fp(wts), fp(zero), fp(S)))),
        ("log-likelihood (pass 4)", lambda: L.oracle_pass_loglik(po.REF_SSE, m, fp(res), fp(P)),
         R and (lambda: R.ref_loglik(m, fp(res), fp(wts), fp(zero), fp(P)))),
        ("J^T W J rank update of given 2x6 rows (part of pass 5)", lambda: L.oracle_rank_update_2x6(fp(J), m, fp(P), po.REF_SSE, A64.ctypes.data_as(C.POINTER(C.c_double))),
         R and (lambda: R.ref_rank_update_2x6(m, fp(J), fp(P), fp(A32)))),
    ]
    print("finest level of S640 (seed 1234): %d selected points, %d valid constraints; best of %d, one thread" % (n, m, REPS))
    print("%-58s %10s %12s" % ("pass", "oracle ms", "reference ms"))
    for name, fo, fr in rows:
        print("%-58s %10.3f %12s" % (name, best(fo), "%.3f" % best(fr) if fr else "-"))
    cfg = po.make_config(3, 0, 100, 5e-7, 0.0, False, mode=po.REF_SSE)
    fr_, fc_ = cm.oracle_pyramids(pair, 4)
    print("whole match (4 levels, 3 -> 0, Precision 5e-7): oracle %.1f ms" % best(lambda: po.match(fr_, fc_, cfg)))


if __name__ == "__main__":
    main()
