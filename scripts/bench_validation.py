#!/usr/bin/env python
"""Loop-closure validation workload on the GPU box: N keyframes of a 640x480 synthetic sweep, the newest
validated against all others (2 proposals each + cross-validation twins; screening at level 3, refinement 3->1), through
the C++ facade -- one device batch per stage vs the reference's one-match-per-proposal pattern -- and through the CPU
oracle (REF_SSE arithmetic, one thread, the reference's own execution model for one validator)."""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from dvo_slam_amd import datagen, tum          # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 33
    from test_validation import build_validator_check, make_voters
    exe = build_validator_check()
    root = tempfile.mkdtemp()
    seq = datagen.synth_sequence(77, n, 640, 480)
    tum.write_dataset(root, seq["grey"], seq["depth"], seq["poses"])
    assoc, gt = os.path.join(root, "assoc.txt"), os.path.join(root, "groundtruth.txt")
    for mode in ("gpu", "gpu_sequential", "gpu", "gpu_sequential"):
        p = subprocess.run([exe, mode, assoc, gt], capture_output=True, text=True, check=True)
        print(p.stderr.strip(), flush=True)
    if "--no-cpu" in sys.argv:
        return
    from oracle import pyoracle as po, validation_oracle as vo
    K = np.array([517.3, 516.5, 318.6, 255.3], np.float32)
    kfs = [vo.Keyframe(k, po.Pyramid(seq["grey"][k].astype(np.float32), po.convert_raw_depth(seq["depth"][k]), K, 4), seq["poses"][k]) for k in range(n)]
    odometry = po.make_config(3, 1, 50, 1e-4, 0.05, True, mode=po.REF_SSE)
    refine = po.make_config(3, 1, 100, 1e-4, 0.05, True, mode=po.REF_SSE)
    screen = po.make_config(3, 3, 100, 1e-4, 0.05, True, mode=po.REF_SSE)
    for k, kf in enumerate(kfs):
        kf.evaluation = vo.LogLikelihoodEvaluation(po.match(kf.image, kfs[k + 1 if k + 1 < n else k - 1].image, odometry, np.eye(4)))
    stages = [vo.Stage(1, screen, False, make_voters("O,N,C0.17,E0.005,X1.0")), vo.Stage(2, refine, True, make_voters("N,C0.17,E0.86"))]
    proposals = []
    for k in range(n - 1):
        proposals += [vo.Proposal.with_identity(kfs[-1], kfs[k]), vo.Proposal.with_relative(kfs[-1], kfs[k])]
    n0 = len(proposals)
    t0 = time.perf_counter()
    out = vo.validate(stages, proposals, lambda stage, p: po.match(p.reference.image, p.current.image, stage.cfg, p.initial))
    print("cpu oracle (REF_SSE, 1 thread) validation: %d proposals (+ %d twins) in %.1f ms, %d accepted" % (n0, n0, 1e3 * (time.perf_counter() - t0), len(out)))


if __name__ == "__main__":
    main()
