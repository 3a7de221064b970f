"""Generates the golden fixtures under tests/golden/ from the CPU oracle (run from the repo root:
`python tests/golden/make_golden.py`).  The reference ships no test vectors, so
these pin OUR oracle: inputs are stored with the expected outputs so that a libm / compiler change that
moved a quantisation boundary in the synthetic generator cannot silently change the fixture.

  s160_seed7.npz   160x120 synthetic pair, levels 2..0, Precision 5e-7: inputs, per-iteration statistics
                   and final transform for both oracle modes (MATH, REF_SSE)
  reduce_kat.npz   the reduce-stage known-answer shape of dvo_core/src/sse_test.cpp:32-102 (random 2x6
                   Jacobian rows, symmetric 2x2 alpha) with the float64 answer
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def flatten(run):
    rows = []
    for L in run["levels"]:
        for it in L["iterations"]:
            rows.append(np.concatenate([[L["id"], it["id"], it["n"], it["neg_ll"]], it["precision"].ravel(), it["x"], it["A"].ravel()]))
    term = np.array([[L["id"], L["valid_pixels"], L["termination"], len(L["iterations"])] for L in run["levels"]], dtype=np.int64)
    return np.array(rows), term


def main():
    w, h, seed, levels = 160, 120, 7, 3
    K = po.FR1_K * (w / 640.0)
    pair = po.synth_pair(seed, w, h, K)
    ref, cur = po.pyramids_from_pair(pair, levels)
    out = dict(grey_ref=pair["grey_ref"], depth_ref=pair["depth_ref"], grey_cur=pair["grey_cur"], depth_cur=pair["depth_cur"],
               K=pair["K"], xi_true=pair["xi_true"], first_level=2, last_level=0, max_iterations=100, precision=5e-7, mu=0.0)
    for name, mode in (("math", po.MATH), ("ref_sse", po.REF_SSE)):
        cfg = po.make_config(first_level=2, last_level=0, max_iterations=100, precision=5e-7, mode=mode)
        run = po.match(ref, cur, cfg)
        rows, term = flatten(run)
        out[name + "_iters"] = rows
        out[name + "_levels"] = term
        out[name + "_T"] = run["T"]
        out[name + "_information"] = run["information"]
        out[name + "_loglik"] = run["loglik"]
    # one linearisation at a non-trivial pose, both passes (w = 1 and t-distribution weights)
    T34 = po.se3_exp(np.array([0.004, -0.003, 0.002, 0.005, -0.004, 0.003]))[:3]
    for name, mode in (("math", po.MATH), ("ref_sse", po.REF_SSE)):
        a = po.level_iteration(ref, cur, 0, T34, first=True, mode=mode)
        b = po.level_iteration(ref, cur, 0, T34, P_prev=a["P"], first=False, mode=mode)
        for tag, r in (("first", a), ("weighted", b)):
            out["lin_%s_%s" % (name, tag)] = np.concatenate([[r["n"], r["neg_ll"]], r["cov"], r["P"].ravel(), r["A"].ravel(), r["b"]])
    out["lin_T34"] = T34
    np.savez_compressed(os.path.join(OUT, "s160_seed7.npz"), **out)

    rng = np.random.default_rng(20130506)
    n = 4096
    J = rng.uniform(-1, 1, size=(n, 12)).astype(np.float32)
    a = rng.uniform(-1, 1, size=(2, 2)).astype(np.float32)
    a[0, 1] = a[1, 0]
    J64 = J.astype(np.float64).reshape(n, 2, 6)
    A = np.einsum("nki,kl,nlj->ij", J64, a.astype(np.float64), J64)
    np.savez_compressed(os.path.join(OUT, "reduce_kat.npz"), J=J, alpha=a, A=A)
    print("wrote fixtures to", OUT)


if __name__ == "__main__":
    main()
