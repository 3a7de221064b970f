"""TUM dataset I/O, evaluation tooling and the replay loop, CPU tier.

The Python module (dvo_slam_amd/tum.py) and the C++ headers (include/dvo_benchmark/) are checked against each other,
against PIL/scipy where present, and the replay loop is run end-to-end with the CPU oracle as the aligner on a small
synthetic sequence written in TUM layout."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from common import ROOT
from dvo_slam_amd import datagen, tum


def build_io_check():
    out = os.path.join(ROOT, "tests", "cpp", "io_check")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "io_check.cpp"), "-o", out, "-lz"])
    return out


def io_check(*args):
    return subprocess.check_output([build_io_check()] + [str(a) for a in args], text=True)


def _filtered_png(path, img, ftype):
    """Encoder used only here: every scanline filtered with `ftype` (0..4), to exercise each decoder branch."""
    img = np.asarray(img)
    arr = img[..., None] if img.ndim == 2 else img
    h, w, ch = arr.shape
    depth = 8 * arr.dtype.itemsize
    raw = np.frombuffer(arr.astype(">u2").tobytes() if depth == 16 else arr.tobytes(), np.uint8).reshape(h, -1).astype(int)
    bpp = ch * depth // 8
    lines = bytearray()
    for y in range(h):
        cur, up = raw[y], raw[y - 1] if y else np.zeros_like(raw[0])
        out = []
        for i in range(len(cur)):
            a = cur[i - bpp] if i >= bpp else 0
            b = up[i]
            c = up[i - bpp] if i >= bpp else 0
            p = a + b - c
            pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
            pred = [0, a, b, (a + b) >> 1, a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)][ftype]
            out.append((cur[i] - pred) & 255)
        lines += bytes([ftype]) + bytes(out)

    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xffffffff)
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[ch]
    comp = zlib.compress(bytes(lines))
    with open(path, "wb") as f:   # two IDAT chunks: the decoder must concatenate them
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) +
                chunk(b"IDAT", comp[:len(comp) // 2]) + chunk(b"IDAT", comp[len(comp) // 2:]) + chunk(b"IEND", b""))


def _checksums(img):
    flat = np.asarray(img).reshape(-1).astype(np.uint64)
    w = 0
    for s in flat.tolist():
        w = (w * 31 + s) % 1000000007
    return int(flat.sum()), w


@pytest.mark.parametrize("ftype", [0, 1, 2, 3, 4])
def test_png_decoders_agree_on_every_filter_type(tmp_path, ftype):
    rng = np.random.default_rng(ftype)
    for k, img in enumerate([rng.integers(0, 256, (9, 13), dtype=np.uint8), rng.integers(0, 65536, (7, 11), dtype=np.uint16),
                             rng.integers(0, 256, (6, 10, 3), dtype=np.uint8), rng.integers(0, 256, (5, 4, 4), dtype=np.uint8)]):
        path = str(tmp_path / ("f%d_%d.png" % (ftype, k)))
        _filtered_png(path, img, ftype)
        got = tum.read_png(path)
        assert got.dtype == img.dtype and got.shape == img.shape and (got == img).all()
        arr = img[..., None] if img.ndim == 2 else img
        w, h, c, bits, s, ws = [int(x) for x in io_check("png", path).split()]
        assert (w, h, c, bits) == (arr.shape[1], arr.shape[0], arr.shape[2], 8 * img.dtype.itemsize)
        assert (s, ws) == _checksums(img)


def test_png_writer_roundtrip_and_pil(tmp_path):
    rng = np.random.default_rng(5)
    for k, img in enumerate([rng.integers(0, 256, (48, 64), dtype=np.uint8), rng.integers(0, 65536, (48, 64), dtype=np.uint16),
                             rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)]):
        path = str(tmp_path / ("w%d.png" % k))
        tum.write_png(path, img)
        assert (tum.read_png(path) == img).all()
        try:
            from PIL import Image
        except ImportError:
            continue
        assert (np.array(Image.open(path)) == img).all()
        Image.fromarray(img).save(str(tmp_path / "pil.png"), optimize=True)   # PIL picks adaptive filters
        assert (tum.read_png(str(tmp_path / "pil.png")) == img).all()
        assert io_check("png", str(tmp_path / "pil.png")).split()[4:] == [str(v) for v in _checksums(img)]


def test_bgr_to_grey_is_opencv_fixed_point():
    # exact grey for grey input, and the BT.601 weights in 14-bit fixed point
    g = np.arange(256, dtype=np.uint8)
    assert (tum.bgr_to_grey(np.stack([g, g, g], -1)) == g).all()
    assert tum.bgr_to_grey(np.array([[[255, 0, 0]]], np.uint8))[0, 0] == (255 * 1868 + 8192) >> 14 == 29
    assert tum.bgr_to_grey(np.array([[[0, 255, 0]]], np.uint8))[0, 0] == 150
    assert tum.bgr_to_grey(np.array([[[0, 0, 255]]], np.uint8))[0, 0] == 76
    rng = np.random.default_rng(1)
    bgr = rng.integers(0, 256, (32, 32, 3), dtype=np.uint8)
    from oracle import pyoracle as po
    assert (po.bgr_to_grey(bgr) == tum.bgr_to_grey(bgr).astype(np.float32)).all()


def _write_small_dataset(root, n=5, w=96, h=72, seed=11):
    seq = datagen.synth_sequence(seed, n, w, h)
    stamps = tum.write_dataset(str(root), seq["grey"], seq["depth"], seq["poses"])
    return seq, stamps


def test_dataset_layout_and_readers(tmp_path):
    seq, stamps = _write_small_dataset(tmp_path)
    assoc = tum.read_associations(str(tmp_path / "assoc.txt"))
    assert len(assoc) == 5 and [a[0] for a in assoc] == pytest.approx(stamps, abs=1e-6)
    # the C++ reader sees the same entries
    lines = io_check("assoc", tmp_path / "assoc.txt").strip().split("\n")
    assert len(lines) == 5
    for line, a in zip(lines, assoc):
        t = line.split()
        assert float(t[0]) == pytest.approx(a[0], abs=1e-6) and t[1] == a[1] and t[3] == a[3]
    # frames come back bit-exact through PNG, both decoders
    for k, a in enumerate(assoc):
        grey, depth = tum.load_frame(str(tmp_path / a[1]), str(tmp_path / a[3]))
        assert (grey == seq["grey"][k]).all() and (depth == seq["depth"][k]).all()
    gs, ds, nan = io_check("frame", tmp_path / assoc[2][1], tmp_path / assoc[2][3]).split()
    g = seq["grey"][2].astype(np.float32).reshape(-1).astype(np.float64)
    from oracle import pyoracle as po
    d = po.convert_raw_depth(seq["depth"][2]).reshape(-1)
    idx = np.arange(g.size)
    assert float(gs) == pytest.approx(float((g * (idx % 97 + 1)).sum()), rel=1e-12)
    assert int(nan) == int(np.isnan(d).sum()) == int((seq["depth"][2] == 0).sum())
    assert float(ds) == pytest.approx(float((np.nan_to_num(d).astype(np.float64) * (idx % 89 + 1)).sum()), rel=1e-12)
    # ground truth round trip + forward-only closest-entry lookup, Python and C++
    gts, gtp = tum.read_trajectory(str(tmp_path / "groundtruth.txt"))
    assert np.abs(gtp - seq["poses"]).max() < 1e-8
    probe = stamps[2] - 0.001
    i = tum.closest_entry(gts, probe)
    assert i == 2
    out = io_check("gt", tmp_path / "groundtruth.txt", "%.6f" % probe).strip().split("\n")
    found, stamp = out[0].split()
    assert int(found) == 1 and float(stamp) == pytest.approx(gts[2], abs=1e-6)
    assert np.abs(np.array(out[1].split(), float).reshape(4, 4) - gtp[2]).max() < 1e-12
    assert np.abs(np.array(out[2].split(), float) - tum.quat_from_rot(gtp[2][:3, :3])).max() < 1e-12
    # past the end: the last entry, not found
    assert tum.closest_entry(gts, gts[-1] + 10) == len(gts) - 1
    assert io_check("gt", tmp_path / "groundtruth.txt", "%.6f" % (gts[-1] + 10)).split()[0] == "0"


def test_sequence_folder_without_an_association_file(tmp_path):
    """A sequence as the data set ships it -- rgb.txt, depth.txt with their own stamps, groundtruth.txt, no assoc.txt: the entries are the
    closest-stamp matches (associate.py's greedy rule, checked against the n x m version), and the folder is found under a root."""
    from oracle import pyoracle as po
    root = tmp_path / "rgbd_dataset_freiburg2_synthetic"
    seq, _ = _write_small_dataset(root, n=6)
    entries = tum.read_associations(str(root / "assoc.txt"))
    os.remove(root / "assoc.txt")
    rng = np.random.default_rng(3)
    depth_stamps = [e[2] + rng.uniform(-0.012, 0.012) for e in entries]
    depth_stamps[4] = entries[4][2] + 0.5                         # a depth frame with no colour frame within 20 ms: dropped
    with open(root / "rgb.txt", "w") as f:
        f.write("# color images\n# file\n# timestamp filename\n")
        f.writelines("%.6f %s\n" % (e[0], e[1]) for e in entries)
    with open(root / "depth.txt", "w") as f:
        f.write("# depth maps\n")
        f.writelines("%.6f %s\n" % (s, e[3]) for s, e in zip(depth_stamps, entries))
    got = tum.sequence_entries(str(root))
    want = tum.associate([e[0] for e in entries], [round(s, 6) for s in depth_stamps])
    assert [(entries[i][1], entries[j][3]) for i, j in want] == [(g[1], g[3]) for g in got] and len(got) == 5
    assert all(abs(g[0] - g[2]) < 0.02 for g in got)
    assert tum.find_sequences(str(tmp_path)) == [str(root)] and tum.find_sequences(str(root)) == [str(root)]
    assert tuple(tum.intrinsics_for(str(root))) == tuple(np.float32(tum.INTRINSICS["freiburg2"]))
    from dvo_slam_amd import replay
    run = replay.replay(str(root), oracle_backend(po.MATH, YAML), str(root / "groundtruth.txt"), K=seq["K"], max_frames=4)
    assert run["failures"] == 0 and len(run["poses"]) == 4 and len(run["relative"]) == 3


def test_quaternion_conventions():
    rng = np.random.default_rng(3)
    for _ in range(50):
        q = rng.normal(size=4)
        T = tum.pose_from_tq(rng.normal(size=3), q)
        assert np.allclose(T[:3, :3] @ T[:3, :3].T, np.eye(3), atol=1e-12) and np.linalg.det(T[:3, :3]) > 0
        q2 = tum.quat_from_rot(T[:3, :3])
        qn = q / np.linalg.norm(q)
        assert np.allclose(q2, qn if qn[3] >= 0 else -qn, atol=1e-12)
    try:
        from scipy.spatial.transform import Rotation
    except ImportError:
        return
    q = rng.normal(size=4)
    assert np.allclose(tum.pose_from_tq([0, 0, 0], q)[:3, :3], Rotation.from_quat(q / np.linalg.norm(q)).as_matrix(), atol=1e-12)


def test_ate_and_rpe_definitions(tmp_path):
    rng = np.random.default_rng(7)
    n = 40
    stamps = 100.0 + np.arange(n) / 30.0
    gt = []
    T = np.eye(4)
    for k in range(n):
        T = T @ tum.pose_from_tq(rng.normal(scale=0.02, size=3), np.r_[rng.normal(scale=0.01, size=3), 1.0])
        gt.append(T.copy())
    gt = np.asarray(gt)
    # an estimate expressed in another world frame has zero ATE and zero RPE
    G = tum.pose_from_tq([0.3, -1.0, 2.0], [0.1, -0.2, 0.3, 0.9])
    est = np.asarray([G @ P for P in gt])
    ate = tum.evaluate_ate(stamps, gt, stamps + 0.001, est)
    assert ate["pairs"] == n and ate["rmse"] < 1e-12
    rpe = tum.evaluate_rpe(gt, est)
    assert rpe["trans_rmse"] < 1e-12 and rpe["rot_rmse"] < 1e-7
    # Horn's closed form is the least-squares optimum: compare with scipy's Kabsch on centred points
    noisy = est.copy()
    noisy[:, :3, 3] += rng.normal(scale=0.01, size=(n, 3))
    ate = tum.evaluate_ate(stamps, gt, stamps, noisy)
    try:
        from scipy.spatial.transform import Rotation
        a, b = noisy[:, :3, 3], gt[:, :3, 3]
        rot, _ = Rotation.align_vectors(b - b.mean(0), a - a.mean(0))
        resid = (rot.apply(a - a.mean(0)) - (b - b.mean(0)))
        assert ate["rmse"] == pytest.approx(np.sqrt((resid ** 2).sum(1).mean()), rel=1e-9)
    except ImportError:
        pass
    assert 0.005 < ate["rmse"] < 0.03 and ate["min"] <= ate["median"] <= ate["max"]
    # a constant per-step translation error of e shows up as RPE = e
    drift = gt.copy()
    for k in range(n):
        drift[k] = gt[k].copy()
    est2 = [gt[0].copy()]
    for k in range(1, n):
        rel = np.linalg.inv(gt[k - 1]) @ gt[k]
        rel[0, 3] += 0.001
        est2.append(est2[-1] @ rel)
    rpe = tum.evaluate_rpe(gt, np.asarray(est2))
    assert rpe["trans_rmse"] == pytest.approx(0.001, rel=1e-9) and rpe["rot_max"] < 1e-7
    # trajectory file round trip and association tolerance
    tum.write_trajectory(str(tmp_path / "est.txt"), stamps, est)
    s2, p2 = tum.read_trajectory(str(tmp_path / "est.txt"))
    assert np.abs(p2 - est).max() < 1e-8 and np.abs(s2 - stamps).max() < 1e-6
    assert len(tum.associate(stamps, stamps + 0.03)) < n and len(tum.associate(stamps, stamps + 0.03, offset=-0.03)) == n


def oracle_backend(mode, cfg_kwargs):
    from oracle import pyoracle as po

    def backend(w, h, K):
        cfg = po.make_config(mode=mode, **cfg_kwargs)
        levels = cfg.first_level + 1

        def make_frame(grey, depth):
            return po.Pyramid(grey.astype(np.float32), po.convert_raw_depth(depth), K, levels)

        def match(ref, cur, T_init):
            r = po.match(ref, cur, cfg, T_init if cfg.use_initial_estimate else None)
            return r["T"] if np.isfinite(r["T"]).all() and np.isfinite(r["information"]).all() else None
        return make_frame, match
    return backend


YAML = dict(first_level=2, last_level=0, max_iterations=50, precision=1e-4, mu=0.05, use_initial_estimate=True)


def test_replay_loop_with_the_oracle_tracks_the_synthetic_sweep(tmp_path):
    """Config 1 in miniature: the benchmark call pattern over a TUM-layout folder with the CPU oracle as dvo_core."""
    from dvo_slam_amd import replay
    from oracle import pyoracle as po
    seq, stamps = _write_small_dataset(tmp_path, n=8, w=160, h=120, seed=21)
    run = replay.replay(str(tmp_path / "assoc.txt"), oracle_backend(po.MATH, YAML), str(tmp_path / "groundtruth.txt"))
    assert run["failures"] == 0 and len(run["poses"]) == 8
    assert np.abs(run["poses"][0] - seq["poses"][0]).max() < 1e-8      # starts at the ground-truth pose
    gts, gtp = tum.read_trajectory(str(tmp_path / "groundtruth.txt"))
    ate = tum.evaluate_ate(gts, gtp, run["stamps"], run["poses"])
    rpe = tum.evaluate_rpe(gtp, run["poses"])
    assert ate["rmse"] < 2e-4 and rpe["trans_rmse"] < 2e-4 and rpe["rot_rmse"] < 2e-4, (ate, rpe)
    # the reference's own arithmetic (REF_SSE quirks) follows the same trajectory to within its approximation noise
    run_sse = replay.replay(str(tmp_path / "assoc.txt"), oracle_backend(po.REF_SSE, YAML), str(tmp_path / "groundtruth.txt"))
    ate_sse = tum.evaluate_ate(gts, gtp, run_sse["stamps"], run_sse["poses"])
    assert abs(ate_sse["rmse"] - ate["rmse"]) < 1e-5, (ate, ate_sse)


def test_long_replay_golden_is_reproducible_from_the_seed():
    """tests/golden/replay_r02.npz (300 noisy 640x480 frames, trajectories of the REFERENCE's own match() and of both oracle
    modes): the generator reproduces the frames of the first steps from the seed, the oracle reproduces its stored relative poses
    bit for bit, and so does the reference's compiled code when oracle/_ref is present."""
    import hashlib
    from dvo_slam_amd import datagen
    from oracle import pyoracle as po
    from common import load_golden
    gold = load_golden("replay_r02.npz")
    seed, n, w, h = [int(v) for v in gold["seq"]]
    k = 4
    seq = datagen.synth_sequence(seed, k, w, h, depth_noise=float(gold["noise"][0]), grey_noise=float(gold["noise"][1]), exposure=float(gold["noise"][2]))
    sums = [int(hashlib.sha1(seq["grey"][i].tobytes() + seq["depth"][i].tobytes()).hexdigest()[:15], 16) for i in range(k)]
    assert sums == gold["checksums"][:k].tolist()
    kw = dict(first_level=3, last_level=1, max_iterations=50, precision=1e-4, mu=0.05, use_initial_estimate=True)
    frames = [(seq["grey"][i].astype(np.float32), po.convert_raw_depth(seq["depth"][i])) for i in range(k)]
    for tag, mode in (("math", po.MATH), ("ref_sse", po.REF_SSE)):
        cfg = po.make_config(mode=mode, **kw)
        pyr = [po.Pyramid(I, Z, gold["K"], 4) for I, Z in frames]
        rel = np.eye(4)
        for i in range(1, k):
            rel = po.match(pyr[i - 1], pyr[i], cfg, rel)["T"]
            assert np.array_equal(rel, gold["benchmark_yaml_%s_relative" % tag][i - 1])
    if po.ref_lib() is not None:
        cfg = po.make_config(**kw)
        rel = np.eye(4)
        for i in range(1, k):
            rel = po.ref_match(frames[i - 1][0], frames[i - 1][1], frames[i][0], frames[i][1], gold["K"], cfg, rel)["T"]
            assert np.array_equal(rel, gold["benchmark_yaml_ref_relative"][i - 1])
    # ref and its restatement agree over all 299 steps of both configurations
    for name in ("benchmark_yaml", "strict_level0"):
        assert np.array_equal(gold[name + "_ref_relative"], gold[name + "_ref_sse_relative"])
