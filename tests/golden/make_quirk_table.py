"""Attributes the distance between the REFERENCE's trajectory and the exact arithmetic's to the reference's quirks, one by one
(run from the repo root: `python tests/golden/make_quirk_table.py`; CPU only, ~10 minutes; writes profiles/r03_quirk_table.txt).

The 300-frame noisy 640x480 sequence of tests/golden/make_replay_golden.py is replayed frame to frame with the CPU oracle in
quirk-selective modes (oracle/dvo_oracle.h: DVO_ORACLE_QUIRKS | bits -- MATH plus exactly the named behaviours of the
reference's SSE path; all bits = REF_SSE = the reference's own match(), bit for bit).  Per mode: the absolute trajectory error
against the true poses, its distance to the reference's, and the largest per-step twist distance to the reference's own relative
poses (tests/golden/replay_r02.npz, computed by the reference's translation units).

  Q1p  _mm_rcp_ps in the projection          dense_tracking_impl.cpp:192      per pixel, order-free: reproducible on a GPU
  Q1w  _mm_rcp_ps in the weights             dense_tracking_impl.cpp:700      per pixel, order-free (groups of four in list order)
  Q2   MXCSR round-toward-zero in the loop   dense_tracking_impl.cpp:165-167  per pixel, order-free
  Q3   odd trailing point dropped            dense_tracking_impl.cpp:169      depends on the compacted list
  Q6   scale pairing bug, float accumulation dense_tracking_impl.cpp:614-615  depends on the ORDER of the compacted list
  Q7   log-likelihood drops n mod 50 terms   dense_tracking_impl.cpp:406-425  depends on the order
  Qf   float sequential normal equations     math_sse.cpp:82-178              depends on the order
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from dvo_slam_amd import datagen, replay, tum  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from golden.make_replay_golden import SEQ, CONFIGS, oracle_backend, frame_checksums  # noqa: E402

Q = po
MODES = [
    ("MATH (what the GPU implements by default)", po.QUIRKS),
    ("MATH + Q1p", po.QUIRKS | Q.Q_RCP_PROJECTION),
    ("MATH + Q1p + Q1w", po.QUIRKS | Q.Q_RCP_PROJECTION | Q.Q_RCP_WEIGHTS),
    ("MATH + Q2", po.QUIRKS | Q.Q_ROUND_TOWARD_ZERO),
    ("MATH + Q1p + Q2", po.QUIRKS | Q.Q_RCP_PROJECTION | Q.Q_ROUND_TOWARD_ZERO),
    ("MATH + Q1p + Q1w + Q2 (every per-pixel quirk)", po.QUIRKS | Q.Q_RCP_PROJECTION | Q.Q_RCP_WEIGHTS | Q.Q_ROUND_TOWARD_ZERO),
    ("MATH + Q6 + Q7", po.QUIRKS | Q.Q_SCALE_PAIRING | Q.Q_LOGLIK_TAIL),
    ("MATH + Q3 + Q6 + Q7 + Qf (every order-dependent quirk)", po.QUIRKS | Q.Q_DROP_ODD | Q.Q_SCALE_PAIRING | Q.Q_LOGLIK_TAIL | Q.Q_FLOAT_NORMAL_EQ),
    # everything except the float SEQUENTIAL rounding of the scale sums: the pairing of Q6 as a formula over the compaction ranks
    # (a GPU could compute the ranks with a scan), accumulated in float64
    ("per-pixel + Q6 as a rank formula (f64)", po.QUIRKS | 1 | 2 | 4 | po.X_PAIRING_F64),
    ("per-pixel + Q6 rank formula + Q7", po.QUIRKS | 1 | 2 | 4 | po.X_PAIRING_F64 | Q.Q_LOGLIK_TAIL),
    ("per-pixel + Q6 rank formula + Q7 + Q3 + Qf", po.QUIRKS | 1 | 2 | 4 | po.X_PAIRING_F64 | Q.Q_LOGLIK_TAIL | Q.Q_DROP_ODD | Q.Q_FLOAT_NORMAL_EQ),
    ("per-pixel + Q3 + Q6 + Q7 (all but Qf)", po.QUIRKS | Q.Q_ALL & ~Q.Q_FLOAT_NORMAL_EQ),
    ("REF_SSE (all)", po.QUIRKS | Q.Q_ALL),
]


def chain(relative):
    poses = [np.eye(4)]
    for T in relative:
        poses.append(poses[-1] @ T)
    return np.asarray(poses)


def main():
    gold = np.load(os.path.join(ROOT, "tests", "golden", "replay_r02.npz"))
    seq = datagen.synth_sequence(SEQ["seed"], SEQ["n"], SEQ["w"], SEQ["h"], depth_noise=SEQ["depth_noise"], grey_noise=SEQ["grey_noise"], exposure=SEQ["exposure"])
    assert frame_checksums(seq).tolist() == gold["checksums"].tolist()
    stamps = np.arange(SEQ["n"], dtype=np.float64)
    lines = []
    for name, kw in CONFIGS.items():
        ref_rel = gold[name + "_ref_relative"]
        ate_ref = tum.evaluate_ate(stamps, seq["poses"], stamps, chain(ref_rel))["rmse"]
        lines.append("## %s  (reference's own trajectory: ATE rmse %.4f mm)" % (name, ate_ref * 1e3))
        lines.append("%-58s %10s %12s %14s %14s" % ("mode", "ATE mm", "vs ref %", "max step dist", "mean step dist"))
        for label, mode in MODES:
            t0 = time.time()
            run = replay.replay_arrays(seq["grey"], seq["depth"], oracle_backend(mode, kw), seq["K"])
            rel = run["relative"]
            ate = tum.evaluate_ate(stamps, seq["poses"], stamps, chain(rel))["rmse"]
            d = np.array([np.abs(po.se3_log(np.linalg.inv(a) @ b)).max() for a, b in zip(rel, ref_rel)])
            lines.append("%-58s %10.4f %+11.2f%% %14.3e %14.3e" % (label, ate * 1e3, 100.0 * (ate - ate_ref) / ate_ref, d.max(), d.mean()))
            print(lines[-1], "  (%.0f s)" % (time.time() - t0), flush=True)
        lines.append("")
    text = __doc__.split("\n\n", 1)[1] + "\n" + "\n".join(lines)
    with open(os.path.join(ROOT, "profiles", "r03_quirk_table.txt"), "w") as f:
        f.write("# Round 3 -- which quirk of the reference separates its trajectory from the exact arithmetic's (tests/golden/make_quirk_table.py)\n" + text)


if __name__ == "__main__":
    main()
