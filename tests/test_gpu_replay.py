"""Sequence replay on the MI355X: the reference's benchmark call pattern over a
TUM-layout folder, through (a) the C++ driver on the facade, (b) the Python mirror of the C-ABI, (c) the CPU oracle."""
import os
import subprocess

import numpy as np
import pytest

from common import ROOT
from test_tum import oracle_backend

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dataset(tmp_path_factory):
    from dvo_slam_amd import datagen, tum
    root = tmp_path_factory.mktemp("seq")
    seq = datagen.synth_sequence(5, 10, 320, 240)
    tum.write_dataset(str(root), seq["grey"], seq["depth"], seq["poses"])
    return root, seq


CONFIGS = {
    # launch/benchmark.yaml of dvo_benchmark (finest level 1), and the strict single-pair setting down to level 0
    "benchmark_yaml": dict(first_level=3, last_level=1, max_iterations=50, precision=1e-4, mu=0.05, use_initial_estimate=True),
    "strict_level0": dict(first_level=3, last_level=0, max_iterations=100, precision=5e-7, mu=0.0, use_initial_estimate=False),
}


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_replay_trajectory_matches_oracle_and_cpp_driver(dataset, name, tmp_path):
    import dvo_slam_amd as d
    from dvo_slam_amd import replay, tum
    from oracle import pyoracle as po
    root, seq = dataset
    kw = CONFIGS[name]
    assoc, gt = str(root / "assoc.txt"), str(root / "groundtruth.txt")
    cfg = d.Config(FirstLevel=kw["first_level"], LastLevel=kw["last_level"], MaxIterationsPerLevel=kw["max_iterations"],
                   Precision=kw["precision"], Mu=kw["mu"], UseInitialEstimate=kw["use_initial_estimate"])
    run_hip = replay.replay(assoc, lambda w, h, K: replay.hip_backend(w, h, K, cfg), gt)
    run_ora = replay.replay(assoc, oracle_backend(po.MATH, kw), gt)
    assert run_hip["failures"] == 0 and run_ora["failures"] == 0
    gts, gtp = tum.read_trajectory(gt)
    ate_hip = tum.evaluate_ate(gts, gtp, run_hip["stamps"], run_hip["poses"])
    ate_ora = tum.evaluate_ate(gts, gtp, run_ora["stamps"], run_ora["poses"])
    print(name, "ATE hip", ate_hip["rmse"], "oracle", ate_ora["rmse"])
    # config 3's acceptance: ATE of the GPU trajectory within 1 % of the oracle's; and pose by pose they coincide
    assert abs(ate_hip["rmse"] - ate_ora["rmse"]) <= 0.01 * ate_ora["rmse"], (ate_hip, ate_ora)
    assert ate_hip["rmse"] < 2e-4
    step = max(np.abs(po.se3_log(np.linalg.inv(a) @ b)).max() for a, b in zip(run_hip["poses"], run_ora["poses"]))
    print(name, "max pose difference (twist inf-norm)", step)
    assert step < 2e-6, step     # nine chained alignments, each stopped at the config's Precision

    # the C++ driver (dvo_slam_amd/apps/benchmark.cpp) prints the same trajectory in TUM format
    exe = os.path.join(ROOT, "dvo_slam_amd", "bin", "dvo_benchmark")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "dvo_slam_amd", "apps"), "-s"])
    out = str(tmp_path / "traj.txt")
    subprocess.check_call([exe, "--rgbdpair_file", assoc, "--groundtruth_file", gt, "--trajectory_file", out,
                           "--coarsest_level", str(kw["first_level"]), "--finest_level", str(kw["last_level"]),
                           "--max_iterations", str(kw["max_iterations"]), "--precision", repr(kw["precision"]), "--mu", repr(kw["mu"]),
                           "--use_initial_estimate", str(int(kw["use_initial_estimate"]))])
    cs, cp = tum.read_trajectory(out)
    assert len(cs) == len(run_hip["stamps"]) and np.abs(cs - run_hip["stamps"]).max() < 1e-6
    assert np.abs(cp - run_hip["poses"]).max() < 1e-12, np.abs(cp - run_hip["poses"]).max()
