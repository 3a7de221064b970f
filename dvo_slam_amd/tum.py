"""TUM RGB-D dataset I/O and trajectory evaluation.

What the reference's benchmark drivers read and write, with the formats they fix:
  * the association list `assoc.txt` -- one line `rgb_stamp rgb_file depth_stamp depth_file` per frame
    (dvo_benchmark/include/dvo_benchmark/rgbd_pair.h:59-71, parsed through file_reader.h:63-99),
  * `groundtruth.txt` -- `stamp tx ty tz qx qy qz qw`, '#' comment lines (groundtruth.h:65-79, tools.h:51-62),
  * 8-bit RGB and 16-bit depth PNGs (depth in 1/5000 m, 0 = no reading: benchmark_slam.cpp:46-93),
  * the estimated trajectory, same 8 columns (benchmark_slam.cpp:490-504, benchmark.cpp:463-478).
Evaluation follows the TUM benchmark tools' definitions: absolute trajectory error after a closed-form rigid alignment
(Horn, no scale) of time-associated positions, and relative pose error over a fixed frame distance.

No OpenCV / PIL here: PNG is decoded and encoded with zlib + numpy (non-interlaced, 8/16-bit grey, RGB, RGBA).
"""
import os
import struct
import zlib

import numpy as np

DEPTH_SCALE = 1.0 / 5000.0   # benchmark_slam.cpp:77


# ---- PNG -------------------------------------------------------------------------------------------------------------
_PNG_MAGIC = b"\x89PNG\r\n\x1a\n"
_CHANNELS = {0: 1, 2: 3, 4: 2, 6: 4}


def _unfilter(raw, height, stride, bpp):
    out = np.zeros((height, stride), np.uint8)
    prev = np.zeros(stride, np.int32)
    pos = 0
    for y in range(height):
        ftype = raw[pos]
        line = np.frombuffer(raw, np.uint8, stride, pos + 1).astype(np.int32)
        pos += stride + 1
        if ftype == 0:
            cur = line
        elif ftype == 1:                      # Sub: running sum per byte lane
            cur = line.copy()
            for c in range(bpp):
                cur[c::bpp] = np.cumsum(line[c::bpp]) & 255
        elif ftype == 2:                      # Up
            cur = (line + prev) & 255
        elif ftype in (3, 4):                 # Average / Paeth: sequential by definition
            cur = np.zeros(stride, np.int32)
            ln, pv = line.tolist(), prev.tolist()
            res = [0] * stride
            for i in range(stride):
                a = res[i - bpp] if i >= bpp else 0
                b = pv[i]
                if ftype == 3:
                    pred = (a + b) >> 1
                else:
                    c = pv[i - bpp] if i >= bpp else 0
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                res[i] = (ln[i] + pred) & 255
            cur = np.asarray(res, np.int32)
        else:
            raise ValueError("png: bad filter type %d" % ftype)
        out[y] = cur
        prev = cur
    return out


def read_png(path):
    """-> array [h, w] or [h, w, c], uint8 or uint16 (native byte order)."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != _PNG_MAGIC:
        raise ValueError("%s: not a PNG file" % path)
    pos, idat, hdr = 8, [], None
    while pos + 8 <= len(data):
        (length,), ctype = struct.unpack(">I", data[pos:pos + 4]), data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + length]
        pos += 12 + length
        if ctype == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif ctype == b"IDAT":
            idat.append(body)
        elif ctype == b"IEND":
            break
    if hdr is None:
        raise ValueError("%s: missing IHDR" % path)
    w, h, depth, ctype, _, _, interlace = hdr
    if interlace or depth not in (8, 16) or ctype not in _CHANNELS:
        raise ValueError("%s: unsupported PNG (bit depth %d, colour type %d, interlace %d)" % (path, depth, ctype, interlace))
    ch = _CHANNELS[ctype]
    bpp = ch * depth // 8
    px = _unfilter(zlib.decompress(b"".join(idat)), h, w * bpp, bpp)
    if depth == 16:
        px = px.reshape(h, w * ch, 2).astype(np.uint16)
        px = (px[..., 0] << 8) | px[..., 1]
    px = px.reshape(h, w, ch)
    return px[..., 0] if ch == 1 else px


def write_png(path, img):
    """img: [h, w] or [h, w, 3|4], uint8 or uint16."""
    img = np.asarray(img)
    if img.dtype not in (np.uint8, np.uint16):
        raise ValueError("write_png: uint8 or uint16 only")
    if img.ndim == 2:
        img = img[..., None]
    h, w, ch = img.shape
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[ch]
    depth = 8 * img.dtype.itemsize
    raw = img.astype(">u2").tobytes() if depth == 16 else img.tobytes()
    stride = w * ch * depth // 8
    lines = b"".join(b"\x00" + raw[y * stride:(y + 1) * stride] for y in range(h))

    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xffffffff)

    with open(path, "wb") as f:
        f.write(_PNG_MAGIC + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(lines, 6)) + chunk(b"IEND", b""))


def bgr_to_grey(bgr):
    """OpenCV's 8-bit CV_BGR2GRAY: fixed-point ITU-R BT.601, (B*1868 + G*9617 + R*4899 + 2^13) >> 14
    (what cv::cvtColor does for the frames at benchmark_slam.cpp:60)."""
    b = bgr[..., 0].astype(np.int32)
    g = bgr[..., 1].astype(np.int32)
    r = bgr[..., 2].astype(np.int32)
    return ((b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14).astype(np.uint8)


def load_frame(rgb_file, depth_file):
    """The `load` of benchmark_slam.cpp:46-93 up to the raw planes: -> (grey u8 [h,w], raw depth u16 [h,w])."""
    rgb = read_png(rgb_file)
    if rgb.dtype != np.uint8:
        raise ValueError("%s: expected an 8-bit colour image" % rgb_file)
    if rgb.ndim == 3:
        grey = bgr_to_grey(rgb[..., 2::-1])           # imread(.., 1) hands OpenCV B,G,R
    else:
        grey = rgb                                   # imread(.., 1) replicates a grey image into 3 equal channels
    depth = read_png(depth_file)
    if depth.ndim != 2 or depth.dtype != np.uint16:
        raise ValueError("%s: expected a 16-bit single-channel depth image" % depth_file)
    return np.ascontiguousarray(grey), np.ascontiguousarray(depth)


# ---- text files ------------------------------------------------------------------------------------------------------
def read_associations(path):
    """assoc.txt -> list of (rgb_stamp, rgb_file, depth_stamp, depth_file); '#' lines skipped (file_reader.h:63-69)."""
    out = []
    with open(path) as f:
        for line in f:
            t = line.split()
            if not t or t[0].startswith("#"):
                continue
            out.append((float(t[0]), t[1], float(t[2]), t[3]))
    return out


def read_file_list(path):
    """rgb.txt / depth.txt of a TUM RGB-D folder -> list of (stamp, file); '#' lines skipped."""
    out = []
    with open(path) as f:
        for line in f:
            t = line.split()
            if len(t) >= 2 and not t[0].startswith("#"):
                out.append((float(t[0]), t[1]))
    return out


def sequence_entries(folder, max_difference=0.02):
    """The association entries of a TUM RGB-D sequence folder: `assoc.txt` / `associations.txt` / `associate.txt` when the folder holds
    one (the file the reference's benchmark reads, benchmark.cpp:402), otherwise rgb.txt and depth.txt matched by closest stamp like the
    data set's associate.py does.  -> list of (rgb_stamp, rgb_file, depth_stamp, depth_file) in time order."""
    for name in ("assoc.txt", "associations.txt", "associate.txt"):
        path = os.path.join(folder, name)
        if os.path.isfile(path):
            return read_associations(path)
    rgb, depth = read_file_list(os.path.join(folder, "rgb.txt")), read_file_list(os.path.join(folder, "depth.txt"))
    ds = np.asarray([s for s, _ in depth])
    cand = []
    for i, (s, _) in enumerate(rgb):                          # the candidates of associate.py, found by bisection instead of n x m
        lo, hi = np.searchsorted(ds, s - max_difference, "left"), np.searchsorted(ds, s + max_difference, "right")
        cand += [(abs(s - ds[j]), i, j) for j in range(lo, hi) if abs(s - ds[j]) < max_difference]
    cand.sort()
    used_rgb, used_depth, out = set(), set(), []
    for _, i, j in cand:
        if i not in used_rgb and j not in used_depth:
            used_rgb.add(i)
            used_depth.add(j)
            out.append((rgb[i][0], rgb[i][1], depth[j][0], depth[j][1]))
    out.sort()
    return out


def find_sequences(root):
    """The sequence folders under `root` (or `root` itself): every folder with a groundtruth.txt and either an association file or
    rgb.txt + depth.txt."""
    def is_sequence(d):
        has = lambda n: os.path.isfile(os.path.join(d, n))
        return has("groundtruth.txt") and (has("assoc.txt") or has("associations.txt") or has("associate.txt") or (has("rgb.txt") and has("depth.txt")))
    if is_sequence(root):
        return [root]
    return sorted(os.path.join(root, d) for d in os.listdir(root) if os.path.isdir(os.path.join(root, d)) and is_sequence(os.path.join(root, d)))


# ROS default calibration of the three Kinect / Xtion sensors of the data set (fx, fy, cx, cy at 640 x 480), keyed by the folder name's
# "freiburg<k>"; the reference's benchmark hard-codes the first (benchmark_slam.cpp:384)
INTRINSICS = {"freiburg1": (517.3, 516.5, 318.6, 255.3), "freiburg2": (520.9, 521.0, 325.1, 249.7), "freiburg3": (535.4, 539.2, 320.1, 247.6)}


def intrinsics_for(folder):
    name = os.path.basename(os.path.normpath(folder))
    for key, K in INTRINSICS.items():
        if key in name:
            return np.asarray(K, np.float32)
    return np.asarray(INTRINSICS["freiburg1"], np.float32)


def read_trajectory(path):
    """groundtruth.txt / estimated trajectory -> (stamps [n], poses [n,4,4])."""
    stamps, poses = [], []
    with open(path) as f:
        for line in f:
            t = line.replace(",", " ").split()
            if len(t) < 8 or t[0].startswith("#"):
                continue
            v = [float(x) for x in t[:8]]
            stamps.append(v[0])
            poses.append(pose_from_tq(v[1:4], v[4:8]))
    return np.asarray(stamps), np.asarray(poses).reshape(-1, 4, 4)


def write_trajectory(path, stamps, poses, header=None):
    """`stamp tx ty tz qx qy qz qw` per line (benchmark_slam.cpp:490-504)."""
    with open(path, "w") as f:
        if header:
            for h in header:
                f.write("# %s\n" % h)
        for s, T in zip(stamps, poses):
            q = quat_from_rot(T[:3, :3])
            f.write("%.6f %.9g %.9g %.9g %.9g %.9g %.9g %.9g\n" % (s, T[0, 3], T[1, 3], T[2, 3], q[0], q[1], q[2], q[3]))


def pose_from_tq(t, q):
    """q = (x, y, z, w) as in the TUM files (tools.h:51-62)."""
    x, y, z, w = np.asarray(q, float) / np.linalg.norm(q)
    T = np.eye(4)
    T[:3, :3] = [[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                 [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                 [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]]
    T[:3, 3] = t
    return T


def quat_from_rot(R):
    """-> (x, y, z, w), w >= 0."""
    R = np.asarray(R, float)
    tr = np.trace(R)
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2
        q = [(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s]
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = [0.0, 0.0, 0.0, (R[k, j] - R[j, k]) / s]
        q[i] = 0.25 * s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
    q = np.asarray(q)
    return -q if q[3] < 0 else q


def closest_entry(stamps, reference, start=0):
    """Index of the first entry whose stamp is >= reference, scanning forward from `start` (tools.h:68-83:
    findClosestEntry only ever moves forward); the last index when the file is exhausted."""
    i = start
    while i < len(stamps) - 1 and stamps[i] < reference:
        i += 1
    return i


# ---- dataset writer (synthetic sequences in TUM layout) ----------------------------------------------------------------
def write_dataset(root, grey, depth, poses, t0=1305031102.175304, rate=30.0):
    """Lay a sequence out like a TUM RGB-D benchmark folder: rgb/<stamp>.png (8-bit RGB), depth/<stamp>.png (16-bit),
    assoc.txt, groundtruth.txt.  -> list of stamps."""
    os.makedirs(os.path.join(root, "rgb"), exist_ok=True)
    os.makedirs(os.path.join(root, "depth"), exist_ok=True)
    stamps = [t0 + k / rate for k in range(len(grey))]
    with open(os.path.join(root, "assoc.txt"), "w") as f:
        for k, s in enumerate(stamps):
            rgb_name, depth_name = "rgb/%.6f.png" % s, "depth/%.6f.png" % s
            write_png(os.path.join(root, rgb_name), np.repeat(grey[k][..., None], 3, axis=2))
            write_png(os.path.join(root, depth_name), depth[k])
            f.write("%.6f %s %.6f %s\n" % (s, rgb_name, s, depth_name))
    write_trajectory(os.path.join(root, "groundtruth.txt"), stamps, poses,
                     header=["ground truth trajectory", "synthetic sequence", "timestamp tx ty tz qx qy qz qw"])
    return stamps


# ---- evaluation ------------------------------------------------------------------------------------------------------
def associate(stamps_a, stamps_b, offset=0.0, max_difference=0.02):
    """Greedy closest-stamp matching of two stamp lists (TUM associate.py semantics) -> list of (ia, ib)."""
    cand = []
    for i, a in enumerate(stamps_a):
        for j, b in enumerate(stamps_b):
            d = abs(a - (b + offset))
            if d < max_difference:
                cand.append((d, i, j))
    cand.sort()
    used_a, used_b, out = set(), set(), []
    for _, i, j in cand:
        if i not in used_a and j not in used_b:
            used_a.add(i)
            used_b.add(j)
            out.append((i, j))
    out.sort()
    return out


def horn_align(model, data):
    """Closed-form rigid alignment (Horn 1987): R, t minimising sum |R model_i + t - data_i|^2.  model, data: [n,3]."""
    model, data = np.asarray(model, float), np.asarray(data, float)
    mc, dc = model.mean(0), data.mean(0)
    W = (model - mc).T @ (data - dc)
    U, _, Vt = np.linalg.svd(W.T)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1
    R = U @ S @ Vt
    return R, dc - R @ mc


def evaluate_ate(gt_stamps, gt_poses, est_stamps, est_poses, offset=0.0, max_difference=0.02):
    """Absolute trajectory error (TUM evaluate_ate): associate by time, align the estimate to the ground truth, report the
    statistics of the translational differences."""
    m = associate(gt_stamps, est_stamps, offset, max_difference)
    if len(m) < 2:
        raise ValueError("evaluate_ate: fewer than two associated poses")
    gt = np.asarray([gt_poses[i][:3, 3] for i, _ in m])
    est = np.asarray([est_poses[j][:3, 3] for _, j in m])
    R, t = horn_align(est, gt)
    err = np.linalg.norm((est @ R.T + t) - gt, axis=1)
    return dict(pairs=len(m), rmse=float(np.sqrt(np.mean(err * err))), mean=float(err.mean()), median=float(np.median(err)),
                std=float(err.std()), min=float(err.min()), max=float(err.max()))


def evaluate_rpe(gt_poses, est_poses, delta=1):
    """Relative pose error over `delta` frames of two index-aligned trajectories: E_i = (Q_i^-1 Q_{i+d})^-1 (P_i^-1 P_{i+d});
    -> translational / rotational RMSE, mean and max (metres, radians)."""
    te, re = [], []
    for i in range(len(gt_poses) - delta):
        dq = np.linalg.inv(gt_poses[i]) @ gt_poses[i + delta]
        dp = np.linalg.inv(est_poses[i]) @ est_poses[i + delta]
        E = np.linalg.inv(dq) @ dp
        te.append(np.linalg.norm(E[:3, 3]))
        re.append(np.arccos(np.clip((np.trace(E[:3, :3]) - 1.0) * 0.5, -1.0, 1.0)))
    te, re = np.asarray(te), np.asarray(re)
    return dict(pairs=len(te), trans_rmse=float(np.sqrt(np.mean(te * te))), trans_mean=float(te.mean()), trans_max=float(te.max()),
                rot_rmse=float(np.sqrt(np.mean(re * re))), rot_mean=float(re.mean()), rot_max=float(re.max()))
