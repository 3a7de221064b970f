"""The drop-in claim, proven with the reference's own code: the reference's tracking front end
(dvo_slam/src/{keyframe_tracker,local_tracker,local_map,tracking_result_evaluation,config}.cpp) and its loop-closure proposal
validation (dvo_slam/src/constraints/*.cpp) are compiled UNMODIFIED against this engine's facade headers (include/dvo/) and
linked with libdvo_hip.so (tests/dropin/Makefile).  The same caller code (oracle/ref_public_api.inc) then runs twice: behind
oracle/_ref it drives the reference's CPU DenseTracker, behind tests/dropin it drives the MI355X engine.  Compared here:
whole matches iteration by iteration, the front end frame by frame, the validator's decisions over ten scenes, and the public
fields of RgbdImage.

Both sides differ by design in the reference's order- / ISA-dependent quirks (approximate rcpps, MXCSR
round-toward-zero, the scale pairing bug, the dropped log-likelihood tail), so the comparison is by tolerance; every bound below is
the measured value with a margin, and the measured values are printed.
"""
import os

import numpy as np
import pytest

import common as cm
from oracle import pyoracle as po

FR1_HALF = (po.FR1_K * 0.5).astype(np.float32)


def need_dropin():
    api = cm.dropin_api()
    if api is None or po.ref_lib() is None:
        pytest.skip("neither the reference tree nor the prebuilt drop-in / reference libraries are present")
    return api


def test_dropin_library_is_the_reference_callers_on_this_engine():
    """CPU tier: the library exists, exports the shared caller entry points and is backed by libdvo_hip (no second tracker inside)."""
    api = need_dropin()
    L = api[0]
    for name in ("dropin_match", "dropin_match_batch", "dropin_validate", "dropin_frontend", "dropin_level_fields"):
        assert hasattr(L, name)
    assert L.dropin_engine().startswith(b"dvo_hip")
    # the reference's own translation units are in there, compiled from /root/reference (symbols of dvo_slam::LocalTracker etc.),
    # and dvo::DenseTracker::match is NOT a symbol of its own: the facade is header-only and forwards to the C-ABI
    import subprocess
    syms = subprocess.check_output(["nm", "-DC", "--defined-only", os.path.join(cm.HERE, "dropin", "_build", "libdvo_dropin.so")], text=True)
    assert "dvo_slam::LocalTracker::update" in syms and "dvo_slam::constraints::ConstraintProposalValidator::validate" in syms
    assert "dvo_slam::KeyframeTracker::update" in syms
    assert "computeResidualsSse" not in syms and "dvo::core::RgbdImage::calculateDerivativeX" not in syms
    undefined = subprocess.check_output(["nm", "-D", "--undefined-only", os.path.join(cm.HERE, "dropin", "_build", "libdvo_dropin.so")], text=True)
    assert "dvo_hip_match" in undefined and "dvo_hip_frame_create_f32" in undefined


def planes_of(pair):
    return (pair["grey_ref"].astype(np.float32), po.convert_raw_depth(pair["depth_ref"]),
            pair["grey_cur"].astype(np.float32), po.convert_raw_depth(pair["depth_cur"]))


@pytest.mark.gpu
@pytest.mark.parametrize("seed,w,h,first,last,iters,precision,mu,use_initial", [
    (1234, 640, 480, 3, 0, 100, 5e-7, 0.0, False),    # BASELINE config 2
    (7, 320, 240, 3, 1, 50, 1e-4, 0.05, True),        # benchmark.yaml
    (11, 640, 480, 3, 1, 50, 1e-4, 0.05, True),
    (21, 640, 480, 3, 3, 100, 5e-7, 0.0, False),      # the validator's screening stage
])
def test_whole_match_every_iteration_against_the_reference(seed, w, h, first, last, iters, precision, mu, use_initial):
    """DenseTracker::match through the facade vs the reference's own match(), record by record: valid-constraint counts, increments
    and log-likelihoods of the common iteration prefix of every level, and the final transform."""
    api = need_dropin()
    pair = cm.synth(seed, w, h)
    cfg = po.make_config(first, last, iters, precision, mu, use_initial)
    r = po.ref_match(*planes_of(pair), pair["K"], cfg)
    g = po.ref_match(*planes_of(pair), pair["K"], cfg, api=api)
    dT = np.abs(po.se3_log(np.linalg.inv(g["T"]) @ r["T"])).max()
    worst_n = worst_x = 0.0
    worst_ll = {}
    assert len(r["levels"]) == len(g["levels"])
    for Lr, Lg in zip(r["levels"], g["levels"]):
        assert Lr["id"] == Lg["id"] and Lr["valid_pixels"] == Lg["valid_pixels"]      # the point selection is exact
        m = min(len(Lr["iterations"]), len(Lg["iterations"]))
        assert m >= 1
        n0 = Lr["iterations"][0]["n"]
        for a, b in zip(Lr["iterations"][:m], Lg["iterations"][:m]):
            worst_n = max(worst_n, abs(a["n"] - b["n"]) / max(n0, 1))
            if np.isfinite(a["x"]).all() and np.isfinite(b["x"]).all():
                worst_x = max(worst_x, np.abs(a["x"] - b["x"]).max())
            # the reference's log-likelihood drops n mod 50 terms and uses the paired scale (Q6, Q7): a few per cent apart
            worst_ll[Lr["id"]] = max(worst_ll.get(Lr["id"], 0.0), abs(a["neg_ll"] - b["neg_ll"]) / abs(a["neg_ll"]))
    print("seed %d %dx%d levels %d..%d: |twist(T_gpu^-1 T_ref)| %.2e, worst |dn|/n %.2e, worst |dx| %.2e, worst relative -ll difference per level %s"
          % (seed, w, h, first, last, dT, worst_n, worst_x, {k: "%.3f" % v for k, v in sorted(worst_ll.items())}))
    # measured: 0.061 / 0.057 / 0.030 / 0.053 on levels 0 / 1 / 2 / 3 (the engine's own restatement, oracle MATH, is held to 1e-6 in
    # test_gpu_parity.py; what is bounded here is the reference's Q6 + Q7 against the exact sum)
    assert all(v <= 0.075 for v in worst_ll.values())
    assert dT < 5e-5 if last < first else dT < 2e-3       # one coarse level alone stops at the coarse level's resolution
    assert worst_n < 0.08                                 # rcpps moves a few of the 100..500 constraints of the 80x60 level across a boundary
    assert worst_x < 5e-3                                 # measured 1e-3 .. 3e-3 on the coarsest level, 1e-5 .. 3e-4 below


@pytest.mark.gpu
def test_reference_tracking_front_end_on_the_engine():
    """The reference's KeyframeTracker -> LocalTracker -> LocalMap, unmodified, on the GPU facade vs on the reference's CPU tracker."""
    api = need_dropin()
    from dvo_slam_amd import datagen
    n = 14
    seq = datagen.synth_sequence(31, n, 320, 240)
    I = [a.astype(np.float32) for a in seq["grey"]]
    Z = [po.convert_raw_depth(a) for a in seq["depth"]]
    cfg = po.make_config(3, 1, 50, 1e-4, 0.05, True)                       # dvo_benchmark/launch/benchmark.yaml
    pr, mr = po.ref_frontend(I, Z, FR1_HALF, cfg, 0.2)                      # the reference's default selection thresholds
    pg, mg = po.ref_frontend(I, Z, FR1_HALF, cfg, 0.2, api=api)
    d = max(np.abs(po.se3_log(np.linalg.inv(a) @ b)).max() for a, b in zip(pr, pg))
    print("completed local maps  reference:", mr.tolist(), " engine:", mg.tolist(), " max pose distance %.2e" % d)
    assert mr.tolist() == mg.tolist() and mr[-1] >= 1
    assert d < 3e-4                                                         # measured 8e-5 over 14 frames
    for k in range(n):                                                      # and both stay on the true trajectory
        assert np.abs(po.se3_log(np.linalg.inv(pg[k]) @ seq["poses"][k])).max() < 3e-3
    # a tight distance threshold puts keyframe decisions on the threshold itself: the two may switch a frame apart, the poses stay close
    pr, mr = po.ref_frontend(I, Z, FR1_HALF, cfg, 0.03)
    pg, mg = po.ref_frontend(I, Z, FR1_HALF, cfg, 0.03, api=api)
    print("tight threshold       reference:", mr.tolist(), " engine:", mg.tolist())
    assert abs(int(mr[-1]) - int(mg[-1])) <= 2
    assert max(np.abs(po.se3_log(np.linalg.inv(a) @ b)).max() for a, b in zip(pr, pg)) < 5e-2


@pytest.mark.gpu
def test_reference_proposal_validator_on_the_engine_over_ten_scenes():
    """The reference's ConstraintProposalValidator + voters, unmodified, on the engine: decision disagreement rate against the same
    code on the reference's CPU tracker, thresholds as KeyframeGraph sets them (not placed in gaps)."""
    api = need_dropin()
    from dvo_slam_amd import datagen
    odometry = po.make_config(3, 1, 50, 1e-4, 0.05, True)
    survivors = differing = 0
    worst = 0.0
    for seed in range(10):
        seq = datagen.synth_sequence(100 + seed, 9, 320, 240)
        I = [a.astype(np.float32) for a in seq["grey"]]
        Z = [po.convert_raw_depth(a) for a in seq["depth"]]
        a = po.ref_validate(I, Z, FR1_HALF, seq["poses"], odometry, 0.17, 0.005, 0.86)
        b = po.ref_validate(I, Z, FR1_HALF, seq["poses"], odometry, 0.17, 0.005, 0.86, api=api)
        sa, sb = [(x["ref"], x["cur"]) for x in a], [(x["ref"], x["cur"]) for x in b]
        survivors += len(set(sa) | set(sb))
        differing += len(set(sa) ^ set(sb))
        for x in a:
            for y in b:
                if (x["ref"], x["cur"]) == (y["ref"], y["cur"]):
                    worst = max(worst, np.abs(po.se3_log(np.linalg.inv(x["T"]) @ y["T"])).max())
    print("validator: %d of %d surviving proposals decided differently over 10 scenes (16 proposals + twins each); "
          "largest distance between common accepted transforms %.2e" % (differing, survivors, worst))
    assert survivors >= 40
    assert differing <= 0.12 * survivors          # measured 3 of 57: decisions that sit on a threshold
    assert worst < 5e-4                           # measured 1.2e-4 (runs stopped at Precision 1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("level", [0, 2])
def test_public_fields_of_rgbd_image_are_host_mirrors(level):
    """RgbdImage::{intensity, depth, intensity_dx, ...} read as FIELDS (as the reference's callers do) after the calls that fill
    them in the reference: bit-identical to the reference's own planes."""
    api = need_dropin()
    pair = cm.synth(5, 320, 240)
    I0, Z0 = pair["grey_ref"].astype(np.float32), po.convert_raw_depth(pair["depth_ref"])
    h, w = 240 >> level, 320 >> level
    planes = np.zeros((6, h, w), np.float32)
    n = api[0].dropin_level_fields(320, 240, po._fp(np.ascontiguousarray(pair["K"], np.float32)), po._fp(I0), po._fp(Z0), level, po._fp(planes))
    assert n == w * h                                                          # point cloud columns
    ref = po.ref_level_planes(I0, Z0, pair["K"], level)["planes"]
    for k in range(6):
        assert np.array_equal(np.isnan(planes[k]), np.isnan(ref[k]))
        assert np.array_equal(np.nan_to_num(planes[k]), np.nan_to_num(ref[k]))


# ---- the reference's BUILT TARGET: dvo_benchmark/src/benchmark_slam.cpp, unmodified, as an executable on this engine --------------
def _golden_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_benchmark_slam_golden", os.path.join(cm.GOLDEN, "make_benchmark_slam_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def tum_folder(tmp_path_factory):
    """The folder the golden trajectories were computed on, regenerated from the seed (frame checksums asserted)."""
    need_dropin()
    mk = _golden_module()
    gold = cm.load_golden("benchmark_slam_r02.npz")
    root = str(tmp_path_factory.mktemp("tum"))
    seq = mk.make_folder(root)
    assert mk.frame_checksums(seq).tolist() == gold["checksums"].tolist(), "the synthetic generator does not reproduce the golden sequence here"
    return mk, gold, root


def _target(name):
    path = os.path.join(cm.HERE, "dropin", "_build", name)
    if not os.path.exists(path):
        pytest.skip("tests/dropin/_build/%s was not built (needs the reference tree at build time)" % name)
    return path


def test_benchmark_slam_binaries_are_the_reference_source_on_either_tracker(tum_folder):
    """CPU tier.  _build/benchmark_slam is benchmark_slam.cpp + the reference's dvo_slam front end on libdvo_hip (no CPU tracker
    inside); _build/benchmark_slam_ref is the same file on the reference's own dvo_core, and reproduces the committed golden."""
    import subprocess
    mk, gold, root = tum_folder
    exe = _target("benchmark_slam")
    syms = subprocess.check_output(["nm", "-C", exe], text=True)
    assert "BenchmarkNode::run()" in syms and "dvo_slam::KeyframeTracker::update" in syms
    assert " U dvo_hip_match" in syms and "computeResidualsSse" not in syms and "calculateDerivativeX" not in syms
    needed = subprocess.check_output(["readelf", "-d", exe], text=True)
    assert "libdvo_hip.so" in needed
    ref = _target("benchmark_slam_ref")
    assert "dvo_hip" not in subprocess.check_output(["nm", "-C", ref], text=True)
    for name, extra in mk.RUNS.items():
        stamps, poses = mk.run_target(ref, root, os.path.join(root, "ref_%s.txt" % name), extra)
        assert np.array_equal(stamps, gold[name + "_stamps"]) and np.array_equal(poses, gold[name + "_poses"]), name


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["defaults", "strict_level0"])
def test_benchmark_slam_built_target_on_the_engine(tum_folder, name):
    """`benchmark_slam _rgbdpair_file:=... _estimate_trajectory:=true ...` -- the reference's executable, its source file compiled
    unmodified against include/dvo/ -- on the MI355X, against the trajectory the same file writes on the reference's CPU tracker
    (golden).  The file prints six significant digits; the front end's keyframe decisions must coincide or poses would jump."""
    from dvo_slam_amd import tum
    mk, gold, root = tum_folder
    exe = _target("benchmark_slam")
    stamps, poses = mk.run_target(exe, root, os.path.join(root, "hip_%s.txt" % name), mk.RUNS[name])
    ref_stamps, ref_poses = gold[name + "_stamps"], gold[name + "_poses"]
    assert np.array_equal(stamps, ref_stamps)          # including the reference reader's duplicate of the last line with stamp 0
    d = max(np.abs(po.se3_log(np.linalg.inv(a) @ b)).max() for a, b in zip(poses, ref_poses))
    gs, gp = tum.read_trajectory(os.path.join(root, "groundtruth.txt"))
    real = stamps > 1.0
    ate_hip = tum.evaluate_ate(gs, gp, stamps[real], poses[real])["rmse"]
    ate_ref = tum.evaluate_ate(gs, gp, ref_stamps[real], ref_poses[real])["rmse"]
    print("benchmark_slam %s: %d poses, largest pose distance to the reference's run %.2e, ATE rmse engine %.6f m reference %.6f m"
          % (name, len(stamps), d, ate_hip, ate_ref))
    assert d < 3e-4, d                # measured 1.4e-4: 40 chained steps of <= 3e-5 each (rcpps) + 6-digit text
    assert abs(ate_hip - ate_ref) < 1e-4 and ate_hip < 1e-3


@pytest.mark.gpu
def test_remaining_public_members_against_the_reference():
    """The members of the public API that callers OUTSIDE match() use -- DenseTracker::computeIntensityErrorImage
    (dense_tracking.cpp:378-444; keyframe_graph.cpp:352-371), operator<< of Config and Stats (dense_tracking.h:218-291;
    keyframe_graph.cpp logs them), IterationStats::InformationEigenValues / InformationConditionNumber
    (dense_tracking_config.cpp:122-135; keyframe_tracker.cpp:170-196) -- through the facade on the MI355X engine against the
    reference's own, same caller code on both sides (oracle/ref_public_api.inc::api_members)."""
    import re
    api = need_dropin()
    L = api[0]
    pair = cm.synth(7, 320, 240)
    cfg = po.make_config(2, 1, 50, 1e-4, 0.05, False)
    T = po.se3_exp(np.array([0.004, -0.003, 0.002, 0.005, -0.004, 0.003]))
    level = 1
    ref = po.ref_api_members(*planes_of(pair), pair["K"], cfg, T, level)
    out = {}
    for compat in (0, 1):
        assert L.dropin_set_engine_option(b"ref_compat", compat) == 0
        out[compat] = po.ref_api_members(*planes_of(pair), pair["K"], cfg, T, level, api=api)
    assert L.dropin_set_engine_option(b"ref_compat", 0) == 0
    # ---- the printers: the same text but for the digits of the floating-point values (Config: identical to the last character) ----
    for compat in (0, 1):
        got, want = out[compat]["text"].splitlines(), ref["text"].splitlines()
        assert got[0] == want[0], (got[0], want[0])                        # operator<<(Config)
        assert len(got) == len(want) or compat == 0, (len(got), len(want))
        skeleton = lambda line: re.sub(r"-?\d+\.?\d*(e[-+]?\d+)?|nan|inf", "#", line)
        for a, b in zip(got, want):
            assert skeleton(a) == skeleton(b), (a, b)
        pixels = lambda lines: [re.search(r"Pixel: (\d+)/(\d+)", x).groups() for x in lines if x.startswith("Level:")]
        assert pixels(got) == pixels(want)                                 # selected / maximum pixels of every level: exact
    print(out[1]["text"])
    # with the reference's reciprocal the iteration structure and the valid-constraint counts agree line by line
    counts = lambda text: [int(m) for m in re.findall(r"ValidConstraints: (\d+)", text)]
    a, b = counts(out[1]["text"]), counts(ref["text"])
    assert len(a) == len(b) and max(abs(x - y) for x, y in zip(a, b)) <= 3, (a, b)
    # ---- eigenvalues / condition number of the information matrix: the facade's own solver against numpy on the same matrix ----
    g = po.ref_match(*planes_of(pair), pair["K"], cfg, api=api)
    info = g["levels"][-1]["iterations"][-1]["A"]
    want_ev = np.linalg.eigvalsh(info)
    assert np.allclose(out[0]["eigenvalues"], want_ev, rtol=1e-9), (out[0]["eigenvalues"], want_ev)
    assert abs(out[0]["condition_number"] - abs(want_ev[-1] / want_ev[0])) <= 1e-9 * out[0]["condition_number"]
    # (against the reference's values only loosely: its precision matrix comes from the pairing scale estimate, SURVEY.md Q6, and the
    # information matrix scales with it -- a few per cent)
    assert np.allclose(ref["eigenvalues"], out[1]["eigenvalues"], rtol=0.15)
    # ---- computeIntensityErrorImage: |intensity residual| of every constraint, 0 elsewhere ----
    e_ref = ref["error_image"]
    # (ref_compat leaves the reference's round-toward-zero mode, SURVEY.md Q2, as the only per-pixel difference: a handful of pixels)
    for compat, frac_tol, val_tol in ((0, 0.02, 5e-3), (1, 0.002, 1e-5)):
        e = out[compat]["error_image"]
        assert e.shape == e_ref.shape == (240 >> level, 320 >> level)
        support_differs = ((e > 0) != (e_ref > 0)).mean()
        both = (e > 0) & (e_ref > 0)
        diff = np.abs(e - e_ref)[both]
        worst = diff.max()
        print("ref_compat %d: constraint masks differ on %.4f %% of the pixels, |difference| on the common ones: largest %.2e, 99.9th percentile %.2e, "
              "median %.2e, above 1e-5: %d of %d (image max %.3f)"
              % (compat, 100 * support_differs, worst, np.percentile(diff, 99.9), np.median(diff), (diff > 1e-5).sum(), diff.size, e_ref.max()))
        assert support_differs <= frac_tol and np.percentile(diff, 99.9 if compat else 99.0) <= val_tol and (diff > 10 * val_tol).mean() <= 2e-3


@pytest.mark.gpu
def test_reference_live_camera_front_end_on_the_engine():
    """dvo_ros/src/camera_dense_tracking.cpp + camera_base.cpp -- the reference's live-camera node (the Affine3d&
    overload of match(), RgbdCameraPyramid::create per frame, SurfacePyramid::convertRawDepthImageSse, configtools.h) -- compiled
    UNMODIFIED against the facade (tests/dropin/Makefile: libdvo_camera_node.so; ROS is stand-ins) and fed eight frames as sensor
    messages: the poses it broadcasts on tf equal the chain of matches made through the Python mirror of the C-ABI on the same planes."""
    import ctypes as C
    import dvo_slam_amd as d
    from dvo_slam_amd import datagen
    path = os.path.join(cm.HERE, "dropin", "_build", "libdvo_camera_node.so")
    if not os.path.exists(path):
        pytest.skip("tests/dropin/_build/libdvo_camera_node.so is not built (needs the reference tree at build time)")
    L = C.CDLL(path)
    n, w, h = 8, 320, 240
    seq = datagen.synth_sequence(9, n, w, h)
    grey = [np.ascontiguousarray(g) for g in seq["grey"]]
    depth_mm = [np.ascontiguousarray(np.round(z.astype(np.float64) / 5.0).astype(np.uint16)) for z in seq["depth"]]      # 1/5000 m -> mm
    K = np.ascontiguousarray(seq["K"], np.float32)
    poses = np.zeros((n, 4, 4))
    vp = C.c_void_p
    L.dropin_camera_node.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(vp), C.POINTER(vp), C.c_int, C.c_int, C.c_int,
                                     C.c_double, C.c_double, C.c_int, C.POINTER(C.c_double)]
    sent = L.dropin_camera_node(n, w, h, K.ctypes.data_as(C.POINTER(C.c_float)), (vp * n)(*[vp(g.ctypes.data) for g in grey]),
                                (vp * n)(*[vp(z.ctypes.data) for z in depth_mm]), 3, 1, 50, 1e-4, 0.05, 0, poses.ctypes.data_as(C.POINTER(C.c_double)))
    assert sent == n - 1                                                 # one transform per frame after the first
    ctx = d.Context(0)
    cam = d.RgbdCameraPyramid(w, h, K, ctx)
    cam.build(4)
    trk = d.DenseTracker(d.Config(FirstLevel=3, LastLevel=1, MaxIterationsPerLevel=50, Precision=1e-4, Mu=0.05, UseInitialEstimate=False), ctx)
    frames = [cam.create(g.astype(np.float32), po.convert_raw_depth(z, 0.001)) for g, z in zip(grey, depth_mm)]
    acc = np.eye(4)
    worst = 0.0
    for k in range(1, n):
        r = d.Result()
        trk.match(frames[k - 1], frames[k], r)
        acc = acc @ r.Transformation
        worst = max(worst, np.abs(poses[k] - acc).max())
    print("live-camera node on the engine, %d frames: largest pose difference to the chain of C-ABI matches %.2e" % (n, worst))
    assert worst < 1e-9
    assert np.abs(po.se3_log(np.linalg.inv(acc) @ np.linalg.inv(seq["poses"][0]) @ seq["poses"][-1])).max() < 5e-3   # ... and it is the motion


# ---- the reference's pose-graph back end, dvo_slam/src/keyframe_graph.cpp, unmodified, inside the reference's executable -------------------
LOOP_SEQ = dict(seed=91, n=24, depth_noise=1.0, grey_noise=3.0, exposure=0.01)
# private parameters of the node (dvo_slam/cfg/dvo_slam.cfg names): keyframes every few centimetres of the 0.1 m sweep, and validation
# thresholds under which some of the loop closures of this short synthetic sweep pass (with the defaults -- tuned for room-sized
# trajectories -- every proposal is voted down on either tracker, which exercises less)
LOOP_ARGS = ["_max_translational_distance:=0.03", "_max_rotational_distance:=0.05", "_constraint_min_entropy_ratio_coarse:=0.005",
             "_constraint_min_entropy_ratio_fine:=0.2", "_constraint_min_eq_sys_constraint_ratio:=0.1", "_graph_opt_robust:=true"]


def _loop_folder(root):
    """A there-and-back camera sweep in TUM layout: 24 frames out, the same 23 frames back (every place is visited twice)."""
    from dvo_slam_amd import datagen, tum
    seq = datagen.synth_sequence(LOOP_SEQ["seed"], LOOP_SEQ["n"], 640, 480, depth_noise=LOOP_SEQ["depth_noise"], grey_noise=LOOP_SEQ["grey_noise"],
                                 exposure=LOOP_SEQ["exposure"])
    idx = list(range(LOOP_SEQ["n"])) + list(range(LOOP_SEQ["n"] - 2, -1, -1))
    tum.write_dataset(root, seq["grey"][idx], seq["depth"][idx], seq["poses"][idx])
    return len(idx)


def _run_graph_target(exe, root, tag, env=None, extra=()):
    """-> (trajectory poses, vertices {id: (stamp, pose7)}, edges {id: (v0, v1, level, chi2, weight, measurement7)}, stderr)"""
    import subprocess
    from dvo_slam_amd import tum
    out = os.path.join(root, "traj_%s.txt" % tag)
    p = subprocess.run([exe, "_rgbdpair_file:=%s/assoc.txt" % root, "_groundtruth_file:=%s/groundtruth.txt" % root, "_estimate_trajectory:=true",
                        "_trajectory_file:=%s" % out] + LOOP_ARGS + list(extra), cwd=root, capture_output=True, text=True, env=dict(os.environ, **(env or {})))
    assert p.returncode == 0, p.stderr[-2000:]
    vertices, edges = {}, {}
    for line in open(os.path.join(root, "assoc_opt_traj_final.txt")):
        t = line.split()
        if t and not t[0].startswith("#"):
            vertices[int(t[8])] = (float(t[0]), np.array([float(x) for x in t[1:8]]))
    for line in open(os.path.join(root, "assoc_error.txt")):
        t = line.split()
        if t and not t[0].startswith("#"):
            edges[int(t[0])] = (int(t[1]), int(t[2]), int(t[3]), float(t[4]), float(t[5]), np.array([float(x) for x in t[6:13]]))
    return tum.read_trajectory(out)[1], vertices, edges, p.stderr


def _keyframes(vertices):
    return sorted(i for i in vertices if i > 0)


def _loop_closures(edges):
    return {i for i, x in edges.items() if x[0] > 0 and x[1] > 0 and abs(x[0] - x[1]) > 1}


def test_keyframe_graph_binaries_are_the_reference_source():
    """CPU tier.  _build/benchmark_slam_graph links dvo_slam/src/keyframe_graph.cpp (KeyframeGraphImpl, its TBB reduction body) and the
    engine; _build/benchmark_slam_graph_ref the same file and the reference's own tracker."""
    import subprocess
    need_dropin()
    exe, ref = _target("benchmark_slam_graph"), _target("benchmark_slam_graph_ref")
    for path in (exe, ref):
        syms = subprocess.check_output(["nm", "-C", path], text=True)
        assert "dvo_slam::internal::KeyframeGraphImpl::execOptimization()" in syms or "KeyframeGraphImpl" in syms
        assert "ValidateConstraintProposalReduction" in syms and "NearestNeighborConstraintSearch::findPossibleConstraints" in syms
    assert " U dvo_hip_match" in subprocess.check_output(["nm", "-C", exe], text=True)
    assert "dvo_hip" not in subprocess.check_output(["nm", "-C", ref], text=True)


@pytest.mark.gpu
def test_reference_keyframe_graph_on_the_engine(tmp_path):
    """The reference's executable with its REAL back end -- dvo_slam/src/keyframe_graph.cpp and keyframe_constraint_search.cpp compiled
    unmodified (proposal generation and the validator pool under tbb::parallel_reduce, keyframe_graph.cpp:500-593, the
    tracker configurations :819-838) -- on a there-and-back sweep, on the MI355X engine against the same executable on the reference's
    CPU tracker, run here.  g2o is a container stand-in whose optimize() is a no-op (oracle/shim/g2o), so the two graphs are compared as
    built: keyframes, odometry edges, and the loop closures the validators let through.

    (1) `use_multithreading` off: the reference is deterministic then, and the graphs must coincide.  (2) on (the node's default): the
    graph thread moves keyframe poses under the tracking thread and the reference's OWN graph changes from run to run (15 or 16
    keyframes, 78 or 91 loop closures on its CPU tracker here), so the engine run -- four validator threads calling
    DenseTracker::match concurrently beside the two of the tracking front end -- is held to what holds for every schedule."""
    need_dropin()
    exe, ref = _target("benchmark_slam_graph"), _target("benchmark_slam_graph_ref")
    root_ref, root_hip = str(tmp_path / "ref"), str(tmp_path / "hip")
    n = _loop_folder(root_ref)
    _loop_folder(root_hip)
    single = ["_use_multithreading:=false"]
    traj_r, vert_r, edge_r, _ = _run_graph_target(ref, root_ref, "ref", extra=single)
    print("reference tracker, one thread: %d frames, %d keyframes, %d graph edges, %d loop closures"
          % (n, len(_keyframes(vert_r)), len(edge_r), len(_loop_closures(edge_r))))
    assert len(_keyframes(vert_r)) >= 8 and len(_loop_closures(edge_r)) >= 10
    for compat in ("0", "1"):
        traj, vert, edge, _ = _run_graph_target(exe, root_hip, "hip" + compat, env={"DVO_HIP_REF_COMPAT": compat}, extra=single)
        common = _loop_closures(edge) & _loop_closures(edge_r)
        def apart(a, b):                                            # twist distance of two edges of one keyframe pair, either direction
            from dvo_slam_amd import tum
            Ta, Tb = tum.pose_from_tq(a[5][:3], a[5][3:]), tum.pose_from_tq(b[5][:3], b[5][3:])
            if (a[0], a[1]) != (b[0], b[1]):                        # the validators kept the twin proposal (current -> reference)
                Tb = np.linalg.inv(Tb)
            return np.abs(po.se3_log(np.linalg.inv(Ta) @ Tb)).max()
        diffs = np.array(sorted(apart(edge[i], edge_r[i]) for i in common))
        worst = diffs[-1]
        print("    measurement differences of the common loop closures: median %.2e, 95 %% %.2e, 99 %% %.2e, the five largest %s"
              % (np.median(diffs), diffs[int(0.95 * len(diffs))], diffs[int(0.99 * len(diffs))], ["%.1e" % x for x in diffs[-5:]]))
        worst_kf = max(np.abs(vert[i][1] - vert_r[i][1]).max() for i in _keyframes(vert_r) if i in vert)
        print("engine, one thread (ref_compat %s): %d keyframes, %d graph edges, %d loop closures, %d in common with the reference's %d; largest "
              "difference of a common loop closure's measurement %.2e, of a keyframe pose %.2e"
              % (compat, len(_keyframes(vert)), len(edge), len(_loop_closures(edge)), len(common), len(_loop_closures(edge_r)), worst, worst_kf))
        assert _keyframes(vert) == _keyframes(vert_r)                 # the front end took the same keyframes
        assert len(common) >= 0.9 * max(len(_loop_closures(edge)), len(_loop_closures(edge_r)))
        assert diffs[int(0.95 * len(diffs))] < 2e-3 and (diffs > 2e-3).mean() < 0.03 and worst_kf < 2e-3
    # (2) the node's default threading on the engine, three times
    for rep in range(3):
        traj, vert, edge, log = _run_graph_target(exe, root_hip, "hipmt%d" % rep)
        loops = _loop_closures(edge)
        assert len(traj) >= n and np.isfinite(traj).all()
        assert 8 <= len(_keyframes(vert)) <= n and len(loops) >= 10
        # every loop closure the validators accepted agrees with the (odometry-chained) vertex estimates to centimetres: chi2 is the
        # information-weighted distance between the measurement and the estimates' relative pose
        off = []
        for i in loops:
            v0, v1 = vert[edge[i][0]][1], vert[edge[i][1]][1]
            off.append(np.linalg.norm(edge[i][5][:3]) - np.linalg.norm(v1[:3] - v0[:3]))
        print("engine, threaded run %d: %d keyframes, %d loop closures; |translation of the measurement| - |distance of its keyframes|: "
              "median %.2e, worst %.2e" % (rep, len(_keyframes(vert)), len(loops), float(np.median(np.abs(off))), float(np.abs(off).max())))
        assert np.abs(off).max() < 0.05                          # (measured 0.017; a diverged alignment on this 0.1 m sweep is off by more)


@pytest.mark.gpu
def test_reference_live_slam_node_on_the_engine(tmp_path):
    """dvo_slam/src/camera_keyframe_tracking.cpp -- the reference's live SLAM node: sensor messages -> KeyframeTracker -> KeyframeGraph,
    reconfigure callbacks, graph publication on every map change, finalOptimization on request -- compiled UNMODIFIED against the facade
    (tests/dropin/Makefile: libdvo_slam_node.so) and fed the there-and-back sweep as messages (mono8 + 32FC1 metres): it tracks and maps
    like the reference's executable benchmark_slam does on the same frames read from PNG files (same engine, same front and back end),
    and its last reconfigure call runs the final optimisation pass over all keyframes."""
    import ctypes as C
    from dvo_slam_amd import datagen
    need_dropin()
    path = os.path.join(cm.HERE, "dropin", "_build", "libdvo_slam_node.so")
    if not os.path.exists(path):
        pytest.skip("tests/dropin/_build/libdvo_slam_node.so is not built (needs the reference tree at build time)")
    exe = _target("benchmark_slam_graph")
    root = str(tmp_path / "loop")
    n = _loop_folder(root)
    # the node hands its FIRST frame to nobody (camera_keyframe_tracking.cpp:252-258: init and return), so the executable gets the folder
    # without it: the same 46 frames then reach the same front end
    with open(os.path.join(root, "assoc.txt")) as f:
        lines = f.readlines()
    with open(os.path.join(root, "assoc.txt"), "w") as f:
        f.writelines(lines[1:])
    stamps_all = [float(l.split()[0]) for l in lines]
    traj_b, vert_b, edge_b, _ = _run_graph_target(exe, root, "bench", extra=["_use_multithreading:=false"])
    seq = datagen.synth_sequence(LOOP_SEQ["seed"], LOOP_SEQ["n"], 640, 480, depth_noise=LOOP_SEQ["depth_noise"], grey_noise=LOOP_SEQ["grey_noise"],
                                 exposure=LOOP_SEQ["exposure"])
    idx = list(range(LOOP_SEQ["n"])) + list(range(LOOP_SEQ["n"] - 2, -1, -1))
    grey = [np.ascontiguousarray(seq["grey"][k]) for k in idx]
    depth = [np.ascontiguousarray(po.convert_raw_depth(seq["depth"][k]), np.float32) for k in idx]
    stamps = np.array(stamps_all)
    K = np.ascontiguousarray(seq["K"], np.float32)
    L = C.CDLL(path)
    vp = C.c_void_p
    L.dropin_slam_node.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_double),
                                   C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int,
                                   C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int,
                                   C.POINTER(C.c_double), C.POINTER(C.c_int)]
    poses = np.zeros((n, 4, 4))
    counts = np.zeros(8, np.int32)
    sent = L.dropin_slam_node(n, 640, 480, K.ctypes.data_as(C.POINTER(C.c_float)), (vp * n)(*[vp(g.ctypes.data) for g in grey]),
                              (vp * n)(*[vp(z.ctypes.data) for z in depth]), stamps.ctypes.data_as(C.POINTER(C.c_double)),
                              3, 1, 50, 1e-4, 0.0, 1, 0.03, 0.05, 0.005, 0.2, 0.1, 1, 0, 1,
                              poses.ctypes.data_as(C.POINTER(C.c_double)), counts.ctypes.data_as(C.POINTER(C.c_int)))
    assert sent == n - 1
    worst = max(np.abs(po.se3_log(np.linalg.inv(a) @ b)).max() for a, b in zip(poses[1:], traj_b[:n - 1]))
    kf_b, loops_b = len(_keyframes(vert_b)), len(_loop_closures(edge_b))
    print("live SLAM node on the engine, %d frames: largest pose distance to benchmark_slam's trajectory file %.2e; map changes %d, keyframes %d, "
          "graph edges %d, loop closures %d (benchmark_slam, which forces a last keyframe: %d keyframes, %d loop closures); after the final "
          "optimisation pass: %d map changes, %d keyframes, %d edges, %d loop closures"
          % (n, worst, counts[0], counts[1], counts[2], counts[3], kf_b, loops_b, counts[4], counts[5], counts[6], counts[7]))
    assert worst < 2e-5                                              # (the file holds six significant digits)
    assert counts[1] in (kf_b, kf_b - 1) and counts[0] >= counts[1] - 1
    assert 0.8 * loops_b <= counts[3] <= loops_b
    assert counts[4] == counts[0] + 1 and counts[5] == counts[1] and counts[7] >= counts[3]
