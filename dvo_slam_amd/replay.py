"""Frame-to-frame visual-odometry replay of a TUM RGB-D association file (SURVEY.md sections 3.1, 8d configs 1 and 3).

The loop of the reference's benchmark driver (dvo_benchmark/src/benchmark.cpp:402-478): align every frame to its
predecessor, feed the last relative motion back as the initial guess (`use_initial_estimate`), chain
`trajectory = trajectory * relative`.  `dvo_slam_amd/apps/benchmark.cpp` is the same loop in C++ over the facade; this
module drives the C-ABI through the Python mirror and is what the evaluation scripts and tests call.

The aligner is injected (`make_frame`, `match`): the product passes the MI355X tracker (`hip_backend`); the tests pass
the CPU oracle through the same loop to compare trajectories.
"""
import os

import numpy as np

from . import tum
from .tracker import Config, DenseTracker, RgbdCameraPyramid, default_context

# launch/benchmark.yaml of dvo_benchmark
BENCHMARK_YAML = dict(FirstLevel=3, LastLevel=1, MaxIterationsPerLevel=50, Precision=1e-4, Mu=0.05, UseInitialEstimate=True)


def hip_backend(width, height, K, cfg=None, ctx=None):
    """-> (make_frame(grey_u8, depth_u16), match(ref, cur, T_init) -> 4x4 or None) on the MI355X engine."""
    ctx = ctx or default_context()
    cfg = cfg or Config(**BENCHMARK_YAML)
    cam = RgbdCameraPyramid(width, height, K, ctx)
    cam.build(cfg.getNumLevels())
    trk = DenseTracker(cfg, ctx)

    def make_frame(grey, depth):
        return cam.create_raw(grey, depth)

    def match(ref, cur, T_init):
        out = trk.match_batch_arrays([ref], [cur], T_init=np.asarray(T_init)[None] if cfg.UseInitialEstimate else None)
        T = out["T"][0]
        return T if np.isfinite(T).all() and np.isfinite(out["information"][0]).all() else None

    return make_frame, match


def replay(assoc_file, backend, groundtruth_file=None, K=None, max_frames=None):
    """-> dict(stamps, poses [n,4,4], failures).  backend(width, height, K) -> (make_frame, match).  `assoc_file` is an association file,
    or a sequence FOLDER (tum.sequence_entries: its association file, or rgb.txt + depth.txt matched by stamp)."""
    if os.path.isdir(assoc_file):
        folder, entries = os.path.abspath(assoc_file), tum.sequence_entries(assoc_file)
    else:
        folder, entries = os.path.dirname(os.path.abspath(assoc_file)), tum.read_associations(assoc_file)
    if not entries:
        raise ValueError("%s: no association entries" % assoc_file)
    if max_frames:
        entries = entries[:max_frames]
    trajectory = np.eye(4)
    if groundtruth_file:
        gs, gp = tum.read_trajectory(groundtruth_file)
        trajectory = gp[tum.closest_entry(gs, entries[0][0])].copy()
    relative = np.eye(4)
    make_frame = match = None
    reference = current = None
    stamps, poses, rel, failures = [], [], [], 0
    for rgb_stamp, rgb_file, _, depth_file in entries:
        grey, depth = tum.load_frame(os.path.join(folder, rgb_file), os.path.join(folder, depth_file))
        if make_frame is None:
            h, w = grey.shape
            k = np.asarray(K if K is not None else np.array([517.3, 516.5, 318.6, 255.3]) * (w / 640.0), np.float32)   # benchmark_slam.cpp:384
            make_frame, match = backend(w, h, k)
        reference, current = current, make_frame(grey, depth)
        if reference is not None:
            T = match(reference, current, relative)
            if T is None:                      # Result::isNaN (Q16)
                failures += 1
                relative = np.eye(4)
            else:
                relative = T
            rel.append(relative.copy())
            trajectory = trajectory @ relative   # benchmark.cpp:463
        stamps.append(rgb_stamp)
        poses.append(trajectory.copy())
    return dict(stamps=np.asarray(stamps), poses=np.asarray(poses), relative=np.asarray(rel), failures=failures)


def replay_arrays(grey, depth, backend, K, stamps=None):
    """The same loop over frames held in memory (grey [n,h,w] u8, depth [n,h,w] u16): no image files in between.  -> dict(stamps,
    poses [n,4,4] with frame 0 at the identity, relative [n-1,4,4], failures)."""
    n, h, w = grey.shape
    make_frame, match = backend(w, h, np.asarray(K, np.float32))
    trajectory, relative = np.eye(4), np.eye(4)
    reference = current = None
    poses, rel, failures = [], [], 0
    for k in range(n):
        reference, current = current, make_frame(grey[k], depth[k])
        if reference is not None:
            T = match(reference, current, relative)
            if T is None:
                failures += 1
                relative = np.eye(4)
            else:
                relative = T
            rel.append(relative.copy())
            trajectory = trajectory @ relative
        poses.append(trajectory.copy())
    return dict(stamps=np.arange(n) / 30.0 if stamps is None else np.asarray(stamps), poses=np.asarray(poses), relative=np.asarray(rel),
                failures=failures)
