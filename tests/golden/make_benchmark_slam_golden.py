"""Generates tests/golden/benchmark_slam_r02.npz (run from the repo root in the BUILD container, where /root/reference is present:
`python tests/golden/make_benchmark_slam_golden.py`).

The reference's built target dvo_benchmark/src/benchmark_slam.cpp, compiled UNMODIFIED and linked with the reference's own
dvo_core (tests/dropin/Makefile: _build/benchmark_slam_ref), is run on a synthetic TUM-layout folder (40 frames of 640x480, fr1
intrinsics as the file hard-codes them, Kinect-model depth noise, intensity noise, exposure drift).  Stored: the trajectory file it
writes, in two configurations (its own defaults; level 0 / strict), and checksums of the frames so that the GPU test -- which runs
the SAME source file linked against this engine's facade (_build/benchmark_slam) on a regenerated folder -- can tell a generator drift
from a tracking difference.
"""
import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dvo_slam_amd import datagen, tum  # noqa: E402

SEQ = dict(seed=77, n=40, w=640, h=480, depth_noise=2.0, grey_noise=4.0, exposure=0.02)
RUNS = {
    "defaults": [],                                                    # dvo_ros/cfg/dvo.cfg + dvo_slam/cfg/dvo_slam.cfg defaults
    "strict_level0": ["_finest_level:=0", "_max_iterations:=100", "_precision:=5e-7", "_use_initial_estimate:=false"],
}


def frame_checksums(seq):
    return np.array([int(hashlib.sha1(seq["grey"][k].tobytes() + seq["depth"][k].tobytes()).hexdigest()[:15], 16) for k in range(len(seq["grey"]))],
                    dtype=np.int64)


def make_folder(root):
    seq = datagen.synth_sequence(SEQ["seed"], SEQ["n"], SEQ["w"], SEQ["h"], depth_noise=SEQ["depth_noise"], grey_noise=SEQ["grey_noise"],
                                 exposure=SEQ["exposure"])
    tum.write_dataset(root, seq["grey"], seq["depth"], seq["poses"])
    return seq


def run_target(exe, root, out, extra):
    """rosrun's own syntax for private parameters (`_name:=value`); -> (stamps, poses [n,4,4]) of the trajectory file written."""
    subprocess.check_call([exe, "_rgbdpair_file:=%s/assoc.txt" % root, "_groundtruth_file:=%s/groundtruth.txt" % root,
                           "_estimate_trajectory:=true", "_trajectory_file:=%s" % out] + list(extra),
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=root)
    return tum.read_trajectory(out)


if __name__ == "__main__":
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "dropin"), "-s"])
    exe = os.path.join(ROOT, "tests", "dropin", "_build", "benchmark_slam_ref")
    store = dict(seq=np.array([SEQ["seed"], SEQ["n"], SEQ["w"], SEQ["h"]]), noise=np.array([SEQ["depth_noise"], SEQ["grey_noise"], SEQ["exposure"]]))
    with tempfile.TemporaryDirectory() as root:
        seq = make_folder(root)
        store["checksums"] = frame_checksums(seq)
        store["poses_true"] = seq["poses"]
        for name, extra in RUNS.items():
            stamps, poses = run_target(exe, root, os.path.join(root, "traj_%s.txt" % name), extra)
            store[name + "_stamps"], store[name + "_poses"] = stamps, poses
            gs, gp = tum.read_trajectory(os.path.join(root, "groundtruth.txt"))
            print(name, len(stamps), "poses; ATE rmse vs ground truth %.6f m" % tum.evaluate_ate(gs, gp, stamps, poses)["rmse"])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "benchmark_slam_r02.npz"), **store)
