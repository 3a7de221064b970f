"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY -- see oracle/dvo_oracle.h.  Importable from tests/, from
__graft_entry__.smoke() and from bench.py's cpu_baseline leg; never from dvo_slam_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

REF_SSE = 0
MATH = 1
# quirk by quirk (dvo_oracle.h): mode = QUIRKS | bits; QUIRKS | Q_ALL == REF_SSE and QUIRKS alone == MATH, bit for bit
QUIRKS = 0x100
Q_RCP_PROJECTION, Q_RCP_WEIGHTS, Q_ROUND_TOWARD_ZERO, Q_DROP_ODD, Q_SCALE_PAIRING, Q_LOGLIK_TAIL, Q_FLOAT_NORMAL_EQ = 1, 2, 4, 8, 16, 32, 64
Q_ALL = 127
X_PAIRING_F64 = 0x200      # experiment (not part of REF_SSE): the pairing of Q6 as a formula over the compaction ranks, float64 sums

TERMINATION = {0: "IterationsExceeded", 1: "IncrementTooSmall", 2: "LogLikelihoodDecreased", 3: "TooFewConstraints", -1: "unset"}


class Config(C.Structure):
    _fields_ = [
        ("first_level", C.c_int32), ("last_level", C.c_int32),
        ("max_iterations_per_level", C.c_int32), ("use_initial_estimate", C.c_int32),
        ("precision", C.c_double), ("mu", C.c_double),
        ("intensity_derivative_threshold", C.c_float), ("depth_derivative_threshold", C.c_float),
        ("mode", C.c_int32), ("reserved", C.c_int32),
    ]


class IterationStats(C.Structure):
    _fields_ = [
        ("id", C.c_int32), ("valid_constraints", C.c_int32),
        ("tdist_loglik", C.c_double), ("tdist_mean", C.c_double * 2), ("tdist_precision", C.c_double * 4),
        ("prior_loglik", C.c_double), ("increment", C.c_double * 6), ("information", C.c_double * 36),
    ]


class LevelStats(C.Structure):
    _fields_ = [("id", C.c_int32), ("max_valid_pixels", C.c_int32), ("valid_pixels", C.c_int32),
                ("termination", C.c_int32), ("n_iterations", C.c_int32), ("first_iteration_index", C.c_int32)]


class Result(C.Structure):
    _fields_ = [("transformation", C.c_double * 16), ("information", C.c_double * 36), ("loglik", C.c_double),
                ("n_levels", C.c_int32), ("n_iterations_total", C.c_int32)]


class IterationOut(C.Structure):
    _fields_ = [("n", C.c_int32), ("n_selected", C.c_int32), ("scale_cov", C.c_float * 3), ("precision", C.c_float * 4),
                ("neg_loglik", C.c_double), ("A", C.c_double * 36), ("b", C.c_double * 6), ("sum_w", C.c_double)]


def build(force=False):
    """Compile liboracle.so with the committed recipe (oracle/Makefile)."""
    if force:
        subprocess.check_call(["make", "-C", _HERE, "-s", "clean"])
    subprocess.check_call(["make", "-C", _HERE, "-s"])      # also builds _ref/libdvo_ref.so when /root/reference is present
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        fp = C.POINTER(C.c_float)
        dp = C.POINTER(C.c_double)
        L.oracle_pyramid_create.restype = C.c_void_p
        L.oracle_pyramid_create.argtypes = [C.c_int, C.c_int, fp, fp, fp, C.c_int]
        L.oracle_pyramid_destroy.argtypes = [C.c_void_p]
        L.oracle_pyramid_num_levels.argtypes = [C.c_void_p]
        L.oracle_pyramid_plane.restype = fp
        L.oracle_pyramid_plane.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), fp]
        L.oracle_pyramid_select.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.POINTER(C.c_uint8)]
        L.oracle_convert_raw_depth.argtypes = [C.POINTER(C.c_uint16), fp, C.c_int, C.c_float]
        L.oracle_bgr_to_grey.argtypes = [C.POINTER(C.c_uint8), fp, C.c_int]
        L.oracle_match.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Config), C.POINTER(Result),
                                   C.POINTER(LevelStats), C.c_int, C.POINTER(IterationStats), C.c_int]
        L.oracle_match_batch.restype = C.c_double
        L.oracle_match_batch.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(Config),
                                         C.POINTER(Result), C.c_int]
        L.oracle_level_iteration.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float,
                                             fp, fp, C.c_int, C.POINTER(IterationOut), fp]
        L.oracle_se3_exp.argtypes = [dp, dp]
        L.oracle_se3_log.argtypes = [dp, dp]
        L.oracle_solve6.argtypes = [dp, dp, dp]
        L.oracle_rank_update_2x6.argtypes = [fp, C.c_int, fp, C.c_int, dp]
        L.oracle_version.restype = C.c_char_p
        L.oracle_pass_residuals.argtypes = [C.c_int, C.c_int, fp, fp, C.c_int, C.c_int, fp, fp, fp, fp]
        L.oracle_pass_weight_vectors.argtypes = [fp, fp, fp]
        L.oracle_pass_weights.argtypes = [C.c_int, C.c_int, fp, fp, fp]
        L.oracle_pass_scale.argtypes = [C.c_int, C.c_int, fp, fp, fp]
        L.oracle_pass_loglik.restype = C.c_double
        L.oracle_pass_loglik.argtypes = [C.c_int, C.c_int, fp, fp]
        _lib = L
    return _lib


REF_LIB_PATH = os.path.join(_HERE, "_ref", "libdvo_ref.so")
_ref = None


def ref_lib():
    """oracle/_ref/libdvo_ref.so: the reference's own SSE passes (see oracle/ref_bridge.cpp), or None when it could not be
    built (no reference tree and no prebuilt library)."""
    global _ref
    if _ref is None:
        build()
        if not os.path.exists(REF_LIB_PATH):
            return None
        L = C.CDLL(REF_LIB_PATH)
        fp = C.POINTER(C.c_float)
        L.ref_compute_residuals.argtypes = [C.c_int, C.c_int, fp, fp, C.c_int, C.c_int, fp, fp, fp, fp, fp, fp]
        L.ref_compute_weights.argtypes = [C.c_int, C.c_int, fp, fp, fp, fp]
        L.ref_compute_scale.argtypes = [C.c_int, C.c_int, fp, fp, fp, fp]
        L.ref_loglik.restype = C.c_float
        L.ref_loglik.argtypes = [C.c_int, fp, fp, fp, fp]
        L.ref_rank_update_2x6.argtypes = [C.c_int, fp, fp, fp]
        L.ref_intrinsics_scale.argtypes = [fp, C.c_float, fp]
        L.ref_convert_raw_depth.argtypes = [C.c_int, C.POINTER(C.c_uint16), C.c_int, C.c_int, C.c_float, fp]
        bind_public_api(L, "ref_")
        L.ref_level_planes.argtypes = [C.c_int, C.c_int, fp, fp, fp, C.c_int, fp, C.POINTER(C.c_uint8), fp, fp]
        _ref = L
    return _ref


def bind_public_api(L, prefix):
    """ctypes signatures of the entry points of oracle/ref_public_api.inc, exported as <prefix>match / match_batch / validate /
    frontend by oracle/_ref (prefix "ref_", the reference's CPU tracker) and by tests/dropin (prefix "dropin_", the reference's
    callers on the MI355X engine)."""
    fp = C.POINTER(C.c_float)
    f = getattr(L, prefix + "match")
    f.argtypes = [C.c_int, C.c_int, fp, fp, fp, fp, fp, C.POINTER(Config), C.POINTER(Result), C.POINTER(LevelStats), C.c_int,
                  C.POINTER(IterationStats), C.c_int]
    f = getattr(L, prefix + "match_batch")
    f.restype = C.c_double
    f.argtypes = [C.c_int, C.c_int, C.c_int, fp, C.POINTER(fp), C.POINTER(fp), C.POINTER(fp), C.POINTER(fp), C.POINTER(Config),
                  C.POINTER(Result), C.c_int, C.c_int]
    f = getattr(L, prefix + "validate")
    f.argtypes = [C.c_int, C.c_int, C.c_int, fp, C.POINTER(fp), C.POINTER(fp), C.POINTER(C.c_double), C.POINTER(Config), C.c_double,
                  C.c_double, C.c_double, C.c_double, C.POINTER(C.c_double), C.c_int]
    f = getattr(L, prefix + "frontend")
    f.argtypes = [C.c_int, C.c_int, C.c_int, fp, C.POINTER(fp), C.POINTER(fp), C.c_void_p, C.c_double, C.c_double, C.c_double,
                  C.POINTER(C.c_double), C.POINTER(C.c_int)]
    f = getattr(L, prefix + "set_engine_option")
    f.argtypes = [C.c_char_p, C.c_int]
    f = getattr(L, prefix + "api_members")
    f.argtypes = [C.c_int, C.c_int, fp, fp, fp, fp, fp, C.POINTER(Config), C.POINTER(C.c_double), C.c_int, fp, C.c_char_p, C.c_int,
                  C.POINTER(C.c_double)]


def _api(api, name):
    """(library, prefix) -> bound function; default = the reference build"""
    L, prefix = api if api else (ref_lib(), "ref_")
    return getattr(L, prefix + name)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


FR1_K = np.array([517.3, 516.5, 318.6, 255.3], dtype=np.float32)   # dvo_benchmark/src/benchmark_slam.cpp:384


def make_config(first_level=3, last_level=1, max_iterations=100, precision=5e-7, mu=0.0, use_initial_estimate=False,
                ithr=0.0, dthr=0.0, mode=MATH):
    """Defaults are dvo_core/src/dense_tracking_config.cpp:27-42."""
    return Config(first_level, last_level, max_iterations, int(use_initial_estimate), precision, mu, ithr, dthr, mode, 0)


class Pyramid:
    def __init__(self, intensity, depth, K=FR1_K, levels=4):
        intensity = np.ascontiguousarray(intensity, dtype=np.float32)
        depth = np.ascontiguousarray(depth, dtype=np.float32)
        assert intensity.shape == depth.shape and intensity.ndim == 2
        K = np.ascontiguousarray(K, dtype=np.float32)
        h, w = intensity.shape
        self.h, self.w, self.levels = h, w, levels
        self.ptr = lib().oracle_pyramid_create(w, h, _fp(K), _fp(intensity), _fp(depth), levels)
        if not self.ptr:
            raise RuntimeError("oracle_pyramid_create failed")

    def __del__(self):
        if getattr(self, "ptr", None):
            lib().oracle_pyramid_destroy(self.ptr)
            self.ptr = None

    def plane(self, level, plane):
        w, h = C.c_int(), C.c_int()
        K = np.zeros(4, np.float32)
        p = lib().oracle_pyramid_plane(self.ptr, level, plane, C.byref(w), C.byref(h), _fp(K))
        if not p:
            raise IndexError((level, plane))
        return np.ctypeslib.as_array(p, shape=(h.value, w.value)).copy(), K

    def select(self, level, ithr=0.0, dthr=0.0):
        w, h = self.w >> level, self.h >> level
        mask = np.zeros((h, w), np.uint8)
        n = lib().oracle_pyramid_select(self.ptr, level, ithr, dthr, mask.ctypes.data_as(C.POINTER(C.c_uint8)))
        return n, mask


def convert_raw_depth(raw, scale=1.0 / 5000.0):
    raw = np.ascontiguousarray(raw, dtype=np.uint16)
    out = np.empty(raw.shape, np.float32)
    lib().oracle_convert_raw_depth(raw.ctypes.data_as(C.POINTER(C.c_uint16)), _fp(out), raw.size, scale)
    return out


def bgr_to_grey(bgr):
    bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
    out = np.empty(bgr.shape[:-1], np.float32)
    lib().oracle_bgr_to_grey(bgr.ctypes.data_as(C.POINTER(C.c_uint8)), _fp(out), out.size)
    return out


def synth_pair(seed, w=640, h=480, K=None):
    """Synthetic pair from the package's data generator (dvo_slam_amd/datagen), fr1 intrinsics scaled to the width."""
    from dvo_slam_amd import datagen
    return datagen.synth_pair(seed, w, h, FR1_K * (w / 640.0) if K is None else K)


def pyramids_from_pair(pair, levels=4):
    ref = Pyramid(pair["grey_ref"].astype(np.float32), convert_raw_depth(pair["depth_ref"]), pair["K"], levels)
    cur = Pyramid(pair["grey_cur"].astype(np.float32), convert_raw_depth(pair["depth_cur"]), pair["K"], levels)
    return ref, cur


def _unpack_stats(res, levels, iters):
    out_levels = []
    for li in range(res.n_levels):
        L = levels[li]
        its = []
        for k in range(L.n_iterations):
            s = iters[L.first_iteration_index + k]
            its.append(dict(id=s.id, n=s.valid_constraints, neg_ll=s.tdist_loglik,
                            precision=np.array(s.tdist_precision).reshape(2, 2), prior_ll=s.prior_loglik,
                            x=np.array(s.increment), A=np.array(s.information).reshape(6, 6)))
        out_levels.append(dict(id=L.id, max_valid_pixels=L.max_valid_pixels, valid_pixels=L.valid_pixels,
                               termination=L.termination, iterations=its))
    return out_levels


def match(ref, cur, cfg, T_init=None):
    """Full coarse-to-fine match. Returns dict(T(4x4), information(6x6), loglik, levels[...])."""
    res = Result()
    T0 = np.eye(4) if T_init is None else np.asarray(T_init, dtype=np.float64)
    for i, v in enumerate(T0.reshape(-1)):
        res.transformation[i] = v
    nl = cfg.first_level - cfg.last_level + 1
    cap_it = nl * (cfg.max_iterations_per_level + 1)
    levels = (LevelStats * nl)()
    iters = (IterationStats * cap_it)()
    rc = lib().oracle_match(ref.ptr, cur.ptr, C.byref(cfg), C.byref(res), levels, nl, iters, cap_it)
    if rc != 0:
        raise RuntimeError("oracle_match rc=%d" % rc)
    return dict(T=np.array(res.transformation).reshape(4, 4), information=np.array(res.information).reshape(6, 6),
                loglik=res.loglik, levels=_unpack_stats(res, levels, iters))


def match_batch(refs, curs, cfg, nthreads=1, T_inits=None):
    n = len(refs)
    results = (Result * n)()
    for i in range(n):
        T0 = np.eye(4) if T_inits is None else np.asarray(T_inits[i], dtype=np.float64)
        for k, v in enumerate(T0.reshape(-1)):
            results[i].transformation[k] = v
    ra = (C.c_void_p * n)(*[r.ptr for r in refs])
    ca = (C.c_void_p * n)(*[c.ptr for c in curs])
    secs = lib().oracle_match_batch(n, ra, ca, C.byref(cfg), results, nthreads)
    Ts = np.stack([np.array(results[i].transformation).reshape(4, 4) for i in range(n)])
    return Ts, secs


def level_iteration(ref, cur, level, T34, P_prev=None, first=True, mode=MATH, ithr=0.0, dthr=0.0, want_residuals=False):
    T34 = np.ascontiguousarray(np.asarray(T34, dtype=np.float32).reshape(-1)[:12])
    Pp = np.zeros(4, np.float32) if P_prev is None else np.ascontiguousarray(np.asarray(P_prev, np.float32).reshape(-1))
    out = IterationOut()
    w, h = ref.w >> level, ref.h >> level
    res = np.empty((h, w, 2), np.float32) if want_residuals else None
    rc = lib().oracle_level_iteration(ref.ptr, cur.ptr, level, mode, ithr, dthr, _fp(T34), _fp(Pp), int(first), C.byref(out),
                                      _fp(res) if want_residuals else None)
    if rc < 0:
        raise RuntimeError("oracle_level_iteration rc=%d" % rc)
    d = dict(rc=rc, n=out.n, n_selected=out.n_selected, cov=np.array(out.scale_cov), P=np.array(out.precision).reshape(2, 2),
             neg_ll=out.neg_loglik, A=np.array(out.A).reshape(6, 6), b=np.array(out.b), sum_w=out.sum_w)
    if want_residuals:
        d["residuals"] = res
    return d


def se3_exp(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    T = np.zeros(16)
    lib().oracle_se3_exp(_dp(x), _dp(T))
    return T.reshape(4, 4)


def se3_log(T):
    T = np.ascontiguousarray(np.asarray(T, dtype=np.float64).reshape(-1))
    x = np.zeros(6)
    lib().oracle_se3_log(_dp(T), _dp(x))
    return x


def solve6(A, b):
    A = np.ascontiguousarray(np.asarray(A, dtype=np.float64).reshape(-1))
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros(6)
    rc = lib().oracle_solve6(_dp(A), _dp(b), _dp(x))
    return x, rc


def rank_update_2x6(J, alpha, mode=MATH):
    J = np.ascontiguousarray(J, dtype=np.float32)
    alpha = np.ascontiguousarray(np.asarray(alpha, dtype=np.float32).reshape(-1))
    A = np.zeros(36)
    lib().oracle_rank_update_2x6(_fp(J), J.shape[0], _fp(alpha), mode, _dp(A))
    return A.reshape(6, 6)


def ref_match(intensity_ref, depth_ref, intensity_cur, depth_cur, K, cfg, T_init=None, api=None):
    """The REFERENCE's DenseTracker::match (oracle/_ref), same dict layout as match().  api = (library, prefix) runs the same
    caller code of another build (tests/dropin)."""
    I0, Z0, I1, Z1 = [np.ascontiguousarray(a, dtype=np.float32) for a in (intensity_ref, depth_ref, intensity_cur, depth_cur)]
    h, w = I0.shape
    K = np.ascontiguousarray(K, dtype=np.float32)
    res = Result()
    T0 = np.eye(4) if T_init is None else np.asarray(T_init, dtype=np.float64)
    for i, v in enumerate(T0.reshape(-1)):
        res.transformation[i] = v
    nl = cfg.first_level - cfg.last_level + 1
    cap_it = nl * (cfg.max_iterations_per_level + 1)
    levels = (LevelStats * nl)()
    iters = (IterationStats * cap_it)()
    rc = _api(api, "match")(w, h, _fp(K), _fp(I0), _fp(Z0), _fp(I1), _fp(Z1), C.byref(cfg), C.byref(res), levels, nl, iters, cap_it)
    if rc != 0:
        raise RuntimeError("ref_match rc=%d" % rc)
    return dict(T=np.array(res.transformation).reshape(4, 4), information=np.array(res.information).reshape(6, 6),
                loglik=res.loglik, levels=_unpack_stats(res, levels, iters))


def ref_api_members(intensity_ref, depth_ref, intensity_cur, depth_cur, K, cfg, T, level, api=None):
    """computeIntensityErrorImage, the printers of Config / Stats and the information eigenvalues of the REFERENCE's public API after
    one match (oracle/ref_public_api.inc::api_members); api = (library, prefix) runs the same caller code on another build."""
    I0, Z0, I1, Z1 = [np.ascontiguousarray(a, dtype=np.float32) for a in (intensity_ref, depth_ref, intensity_cur, depth_cur)]
    h, w = I0.shape
    K = np.ascontiguousarray(K, dtype=np.float32)
    T = np.ascontiguousarray(T, dtype=np.float64)
    err = np.zeros((h >> level, w >> level), np.float32)
    text = C.create_string_buffer(1 << 16)
    eig = (C.c_double * 7)()
    rc = _api(api, "api_members")(w, h, _fp(K), _fp(I0), _fp(Z0), _fp(I1), _fp(Z1), C.byref(cfg), _dp(T), level, _fp(err), text, len(text), eig)
    assert rc == err.shape[0] * 65536 + err.shape[1], (rc, err.shape)
    return dict(error_image=err, text=text.value.decode(), eigenvalues=np.array(eig[:6]), condition_number=eig[6])


def ref_level_planes(intensity, depth, K, level, want_points=False):
    """Planes [6,h,w], selection mask, level intrinsics (and the selected points [n,12]) of the REFERENCE's image model."""
    L = ref_lib()
    I, Z = np.ascontiguousarray(intensity, np.float32), np.ascontiguousarray(depth, np.float32)
    h, w = I.shape
    lh, lw = h >> level, w >> level
    planes = np.zeros((6, lh, lw), np.float32)
    mask = np.zeros((lh, lw), np.uint8)
    Kl = np.zeros(4, np.float32)
    pts = np.zeros((lh * lw, 12), np.float32) if want_points else None
    K = np.ascontiguousarray(K, dtype=np.float32)
    n = L.ref_level_planes(w, h, _fp(K), _fp(I), _fp(Z), level, _fp(planes), mask.ctypes.data_as(C.POINTER(C.c_uint8)), _fp(Kl),
                           _fp(pts) if want_points else None)
    return dict(planes=planes, mask=mask, K=Kl, n_selected=n, points=None if pts is None else pts[:n])


def ref_match_batch(planes, K, cfg, n_matches=None, nthreads=1, T_inits=None, api=None):
    """Throughput of the REFERENCE's DenseTracker::match (oracle/_ref): planes = list of (I_ref, Z_ref, I_cur, Z_cur) float32
    arrays; n_matches >= len(planes) matches are run round-robin on `nthreads` threads.  -> (T [n,4,4], seconds)."""
    n = len(planes)
    n_matches = n_matches or n
    keep = [[np.ascontiguousarray(p[k], dtype=np.float32) for p in planes] for k in range(4)]
    h, w = keep[0][0].shape
    fp = C.POINTER(C.c_float)
    arrs = [(fp * n)(*[_fp(a) for a in keep[k]]) for k in range(4)]
    results = (Result * n)()
    for i in range(n):
        T0 = np.eye(4) if T_inits is None else np.asarray(T_inits[i], dtype=np.float64)
        for k, v in enumerate(T0.reshape(-1)):
            results[i].transformation[k] = v
    K = np.ascontiguousarray(K, dtype=np.float32)
    secs = _api(api, "match_batch")(n, w, h, _fp(K), arrs[0], arrs[1], arrs[2], arrs[3], C.byref(cfg), results, n_matches, nthreads)
    return np.stack([np.array(results[i].transformation).reshape(4, 4) for i in range(n)]), secs


def ref_validate(intensity, depth, K, poses, odometry_cfg, min_constraint_ratio, min_entropy_coarse, min_entropy_fine, cross_threshold=1.0, api=None):
    """The REFERENCE's ConstraintProposalValidator (oracle/_ref, see ref_bridge.cpp::ref_validate): the last keyframe against all
    others.  -> list of dict(ref, cur, score, T) for the surviving proposals."""
    n = len(intensity)
    keepI = [np.ascontiguousarray(a, np.float32) for a in intensity]
    keepZ = [np.ascontiguousarray(a, np.float32) for a in depth]
    h, w = keepI[0].shape
    fp = C.POINTER(C.c_float)
    I = (fp * n)(*[_fp(a) for a in keepI])
    Z = (fp * n)(*[_fp(a) for a in keepZ])
    P = np.ascontiguousarray(np.asarray(poses, np.float64).reshape(n, 16))
    K = np.ascontiguousarray(K, dtype=np.float32)
    out = np.zeros((4 * n, 19))
    m = _api(api, "validate")(n, w, h, _fp(K), I, Z, P.ctypes.data_as(C.POINTER(C.c_double)), C.byref(odometry_cfg), min_constraint_ratio,
                       min_entropy_coarse, min_entropy_fine, cross_threshold, out.ctypes.data_as(C.POINTER(C.c_double)), 4 * n)
    return [dict(ref=int(o[0]), cur=int(o[1]), score=float(o[2]), T=o[3:].reshape(4, 4).copy()) for o in out[:m]]


def ref_frontend(intensity, depth, K, tracking_cfg, max_translational_distance=0.2, min_entropy_ratio=0.91, min_constraint_ratio=0.33, api=None):
    """The REFERENCE's tracking front end (oracle/_ref, see ref_bridge.cpp::ref_frontend): KeyframeTracker::update() frame by frame.
    -> (poses [n,4,4], completed local maps after each frame [n])."""
    n = len(intensity)
    keepI = [np.ascontiguousarray(a, np.float32) for a in intensity]
    keepZ = [np.ascontiguousarray(a, np.float32) for a in depth]
    h, w = keepI[0].shape
    fp = C.POINTER(C.c_float)
    I = (fp * n)(*[_fp(a) for a in keepI])
    Z = (fp * n)(*[_fp(a) for a in keepZ])
    K = np.ascontiguousarray(K, dtype=np.float32)
    poses = np.zeros((n, 16))
    maps = np.zeros(n, np.int32)
    _api(api, "frontend")(n, w, h, _fp(K), I, Z, C.cast(C.byref(tracking_cfg), C.c_void_p), max_translational_distance, min_entropy_ratio,
                   min_constraint_ratio, poses.ctypes.data_as(C.POINTER(C.c_double)), maps.ctypes.data_as(C.POINTER(C.c_int)))
    return poses.reshape(n, 4, 4), maps
