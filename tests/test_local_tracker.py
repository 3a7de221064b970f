"""The tracking front-end pattern: keyframe and previous-frame alignment of every new frame as one device
batch (include/dvo_slam/local_tracker.h) against the sequential CPU restatement (oracle/frontend_oracle.py)."""
import os
import subprocess

import numpy as np
import pytest

from common import ROOT


def build_local_tracker_check():
    import dvo_slam_amd as d
    d.build()
    out = os.path.join(ROOT, "tests", "cpp", "local_tracker_check")
    libdir = os.path.join(ROOT, "dvo_slam_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "local_tracker_check.cpp"), "-o", out, "-L" + libdir, "-ldvo_hip",
                           "-Wl,-rpath," + libdir, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lz"])
    return out


def test_front_end_headers_compile_with_the_host_compiler():
    assert os.path.exists(build_local_tracker_check())


@pytest.mark.gpu
def test_local_tracker_follows_the_sequential_oracle(tmp_path):
    from dvo_slam_amd import datagen, tum
    from oracle import frontend_oracle as fo, pyoracle as po
    n, max_distance = 14, 0.03
    seq = datagen.synth_sequence(31, n, 320, 240)
    tum.write_dataset(str(tmp_path), seq["grey"], seq["depth"], seq["poses"])
    out = subprocess.check_output([build_local_tracker_check(), str(tmp_path / "assoc.txt"), repr(max_distance)], text=True)
    got = [(int(line.split()[0]), np.array(line.split()[1:], float).reshape(4, 4)) for line in out.strip().split("\n")]
    assert len(got) == n - 1

    K = (np.array([517.3, 516.5, 318.6, 255.3]) * 0.5).astype(np.float32)
    frames = [po.Pyramid(seq["grey"][k].astype(np.float32), po.convert_raw_depth(seq["depth"][k]), K, 4) for k in range(n)]
    cfg = po.make_config(3, 1, 50, 1e-4, 0.05, True, mode=po.MATH)
    asked = []
    lt = fo.LocalTracker(lambda ref, cur, T0: po.match(ref, cur, cfg, T0),
                         [lambda ro, rk: bool(np.linalg.norm(rk["T"][:3, 3]) < max_distance), lambda ro, rk: asked.append(1) is None])
    lt.init_new_local_map(frames[0], frames[1])
    want = [(False, lt.current_pose.copy())] + [lt.update(frames[k])[::-1] for k in range(2, n)]
    switches = [int(s) for s, _ in want]
    print("keyframe switches:", switches)
    assert 2 <= sum(switches) <= n - 4, "the scenario should open a few new local maps, not one per frame"
    assert len(asked) == n - 2
    assert [s for s, _ in got] == switches
    for (_, Tg), (_, Tw) in zip(got, want):
        assert np.abs(po.se3_log(np.linalg.inv(Tg) @ Tw)).max() < 2e-6
    # and the chained estimate stays on the true trajectory (frame 0 is the world frame)
    for k, (_, Tg) in enumerate(got, start=1):
        assert np.abs(po.se3_log(np.linalg.inv(Tg) @ seq["poses"][k])).max() < 2e-3


@pytest.mark.gpu
@pytest.mark.parametrize("max_distance,min_ratio_note", [(0.04, "distance and quality both bite"), (0.012, "divergence: 1.5 x distance is exceeded")])
def test_keyframe_selection_criteria_follow_the_oracle(tmp_path, max_distance, min_ratio_note):
    """LocalTracker driven by the KeyframeTracker criteria (include/dvo_slam/keyframe_selection.h) vs their restatement."""
    from dvo_slam_amd import datagen, tum
    from oracle import frontend_oracle as fo, pyoracle as po
    n = 16
    seq = datagen.synth_sequence(31, n, 320, 240)
    tum.write_dataset(str(tmp_path), seq["grey"], seq["depth"], seq["poses"])
    out = subprocess.check_output([build_local_tracker_check(), str(tmp_path / "assoc.txt"), repr(max_distance), "selection"], text=True)
    got = [(int(line.split()[0]), np.array(line.split()[1:], float).reshape(4, 4)) for line in out.strip().split("\n")]
    K = (np.array([517.3, 516.5, 318.6, 255.3]) * 0.5).astype(np.float32)
    frames = [po.Pyramid(seq["grey"][k].astype(np.float32), po.convert_raw_depth(seq["depth"][k]), K, 4) for k in range(n)]
    cfg = po.make_config(3, 1, 50, 1e-4, 0.05, True, mode=po.MATH)
    sel = fo.KeyframeSelection(max_translational_distance=max_distance)
    lt = fo.LocalTracker(lambda ref, cur, T0: po.match(ref, cur, cfg, T0), sel.callbacks(), sel.on_map_initialized)
    lt.init_new_local_map(frames[0], frames[1])
    want = [(False, lt.current_pose.copy())] + [lt.update(frames[k])[::-1] for k in range(2, n)]
    switches = [int(s) for s, _ in want]
    print(min_ratio_note, "switches:", switches, "quality ratios:", np.round(sel.trace, 3))
    assert sum(switches) >= 1
    # no decision of this scenario sits on a threshold (the comparison below would otherwise be fragile)
    assert min(abs(r - sel.min_ratio) for r in sel.trace) > 2e-3
    assert [s for s, _ in got] == switches
    for (_, Tg), (_, Tw) in zip(got, want):
        assert np.abs(po.se3_log(np.linalg.inv(Tg) @ Tw)).max() < 2e-6
