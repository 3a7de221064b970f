"""Host-side mirror of the reference's dvo_core class API for the alignment hot path.

Same names, argument meaning and error behaviour as
  dvo::DenseTracker / Config / Result / Stats      dvo_core/include/dvo/dense_tracking.h:36-215
  dvo::core::RgbdCameraPyramid / RgbdImagePyramid  dvo_core/include/dvo/core/rgbd_image.h:99-262
  dvo::core::PointSelection                         dvo_core/include/dvo/core/point_selection.h:69-99
but every pixel operation runs on the MI355X through libdvo_hip.so (include/dvo_hip.h).  The
C++ facade with the identical role for C++ callers lives in include/dvo/.
"""
import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from ._lib import DvoHipError

TERMINATION = {0: "IterationsExceeded", 1: "IncrementTooSmall", 2: "LogLikelihoodDecreased", 3: "TooFewConstraints",
               -1: "unset"}


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class Context:
    """One device + one HIP stream + scratch (dvo_hip_context).  One per host thread."""

    def __init__(self, device=0):
        self._lib = _lib.lib()
        self.ptr = C.c_void_p()
        rc = self._lib.dvo_hip_context_create(device, C.byref(self.ptr))
        if rc != _lib.OK:
            raise DvoHipError(rc, self._lib.dvo_hip_last_error(None).decode())
        self.device = device

    def check(self, rc):
        if rc != _lib.OK:
            raise DvoHipError(rc, self._lib.dvo_hip_last_error(self.ptr).decode())

    def set_option(self, key, value):
        self.check(self._lib.dvo_hip_set_option(self.ptr, key.encode(), int(value)))

    def counter(self, key):
        """An event counter of the context ("resident_launches", "resident_timeouts")."""
        v = C.c_longlong(0)
        self.check(self._lib.dvo_hip_get_counter(self.ptr, key.encode(), C.byref(v)))
        return v.value

    @property
    def stream(self):
        return self._lib.dvo_hip_context_stream(self.ptr)

    def close(self):
        if getattr(self, "ptr", None):
            self._lib.dvo_hip_context_destroy(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_context = None


def default_context():
    global _default_context
    if _default_context is None:
        _default_context = Context(0)
    return _default_context


@dataclass
class Config:
    """dvo::DenseTracker::Config; defaults = dvo_core/src/dense_tracking_config.cpp:27-42."""
    FirstLevel: int = 3
    LastLevel: int = 1
    MaxIterationsPerLevel: int = 100
    Precision: float = 5e-7
    Mu: float = 0.0
    UseInitialEstimate: bool = False
    UseWeighting: bool = True            # dead for match(); kept so callers compile
    UseParallel: bool = False            # dead
    InfluenceFuntionType: int = 2        # TDistribution (dead)
    InfluenceFunctionParam: float = 5.0  # dead
    ScaleEstimatorType: int = 2          # TDistribution (dead)
    ScaleEstimatorParam: float = 5.0     # dead
    IntensityDerivativeThreshold: float = 0.0
    DepthDerivativeThreshold: float = 0.0

    def getNumLevels(self):
        return self.FirstLevel + 1

    def UseEstimateSmoothing(self):
        return self.Mu > 1e-6

    def IsSane(self):
        return self.FirstLevel >= self.LastLevel

    def to_c(self):
        return _lib.Config(self.FirstLevel, self.LastLevel, self.MaxIterationsPerLevel, int(self.UseInitialEstimate),
                           self.Precision, self.Mu, self.IntensityDerivativeThreshold, self.DepthDerivativeThreshold)


@dataclass
class IterationStats:
    Id: int = 0
    ValidConstraints: int = 0
    TDistributionLogLikelihood: float = 0.0
    TDistributionMean: np.ndarray = None
    TDistributionPrecision: np.ndarray = None
    PriorLogLikelihood: float = 0.0
    EstimateIncrement: np.ndarray = None
    EstimateInformation: np.ndarray = None

    def InformationEigenValues(self):
        return np.sort(np.linalg.eigvals(self.EstimateInformation).real)

    def InformationConditionNumber(self):
        ev = self.InformationEigenValues()
        return abs(ev[5] / ev[0])


@dataclass
class LevelStats:
    Id: int = 0
    MaxValidPixels: int = 0
    ValidPixels: int = 0
    TerminationCriterion: int = -1
    Iterations: list = field(default_factory=list)

    def HasIterationWithIncrement(self):   # dense_tracking_config.cpp:138-143
        need = 2 if self.TerminationCriterion in (2, 3) else 1
        return len(self.Iterations) >= need

    def LastIterationWithIncrement(self):
        assert self.HasIterationWithIncrement()
        return self.Iterations[-2] if self.TerminationCriterion == 2 else self.Iterations[-1]

    def LastIteration(self):
        return self.Iterations[-1]


@dataclass
class Stats:
    Levels: list = field(default_factory=list)


class Result:
    """dvo::DenseTracker::Result (dense_tracking.h:125-140, dense_tracking_config.cpp:96-121)."""

    def __init__(self):
        self.Transformation = np.full((4, 4), np.nan)
        self.Transformation[3] = [0, 0, 0, 1]
        self.Information = np.eye(6)
        self.LogLikelihood = np.finfo(np.float64).max
        self.Statistics = Stats()
        self.Entropy = self.ConditionNumber = self.ConstraintRatio = float("nan")
        self.ConstraintRatioAccepted = 0.0

    def isNaN(self):
        return not (np.isfinite(self.Transformation.sum()) and np.isfinite(self.Information.sum()))

    def setIdentity(self):
        self.Transformation = np.eye(4)
        self.Information = np.eye(6)
        self.LogLikelihood = 0.0

    def clearStatistics(self):
        self.Statistics.Levels = []


class RgbdImage:
    """One pyramid level: host mirrors of the device planes, downloaded lazily (rgbd_image.h:161-179)."""
    _PLANES = {"intensity": 0, "depth": 1, "intensity_dx": 2, "intensity_dy": 3, "depth_dx": 4, "depth_dy": 5}

    def __init__(self, pyramid, level):
        self._pyr, self._level = pyramid, level
        w, h = C.c_int(), C.c_int()
        K = np.zeros(4, np.float32)
        pyramid.ctx.check(pyramid.ctx._lib.dvo_hip_frame_info(pyramid.ptr, level, C.byref(w), C.byref(h), _fp(K)))
        self.width, self.height, self.K = w.value, h.value, K
        self.timestamp = pyramid._timestamp
        self._cache = {}

    def __getattr__(self, name):
        planes = type(self)._PLANES
        if name in planes:
            if name not in self._cache:
                out = np.empty((self.height, self.width), np.float32)
                ctx = self._pyr.ctx
                ctx.check(ctx._lib.dvo_hip_frame_download_plane(ctx.ptr, self._pyr.ptr, self._level, planes[name], _fp(out)))
                self._cache[name] = out
            return self._cache[name]
        raise AttributeError(name)

    def buildPointCloud(self):        # device side needs nothing: points are recomputed from Z
        pass

    def buildAccelerationStructure(self):   # built at frame creation
        pass


class RgbdImagePyramid:
    def __init__(self, camera, make_frame, levels, timestamp=0.0):
        self.camera, self.ctx = camera, camera.ctx
        self._make_frame = make_frame
        self._timestamp = timestamp
        self.levels = 0
        self.ptr = None
        self.build(levels)

    def build(self, num_levels):            # rgbd_image.cpp:156-172 (idempotent, only ever grows)
        if self.levels >= num_levels:
            return
        if self.ptr:
            self.ctx._lib.dvo_hip_frame_destroy(self.ctx.ptr, self.ptr)
        self.ptr = self._make_frame(num_levels)
        self.levels = num_levels

    compute = build                         # deprecated alias in the reference too

    def update_raw_device(self, grey_dev_ptr, depth_dev_ptr, depth_scale=1.0 / 5000.0):
        """Re-ingest new raw planes already in HBM into this pyramid (no allocation, asynchronous)."""
        import ctypes as C
        self.ctx.check(self.ctx._lib.dvo_hip_frame_update_raw_device(self.ctx.ptr, self.ptr, C.c_void_p(grey_dev_ptr),
                                                                     C.c_void_p(depth_dev_ptr), depth_scale))

    def level(self, idx):
        assert idx < self.levels
        return RgbdImage(self, idx)

    def timestamp(self):
        return self._timestamp

    def __del__(self):
        try:
            if self.ptr and self.ctx.ptr:
                self.ctx._lib.dvo_hip_frame_destroy(self.ctx.ptr, self.ptr)
        except Exception:
            pass
        self.ptr = None


class RgbdCameraPyramid:
    """RgbdCameraPyramid(width, height, intrinsics) with intrinsics = (fx, fy, ox, oy)."""

    def __init__(self, base_width, base_height, base_intrinsics, ctx=None):
        self.ctx = ctx or default_context()
        self.width, self.height = int(base_width), int(base_height)
        self.K = np.ascontiguousarray(base_intrinsics, dtype=np.float32)
        assert self.K.shape == (4,)
        self.levels = 1

    def build(self, levels):                # rgbd_image.cpp:283-296
        self.levels = max(self.levels, int(levels))

    def create(self, base_intensity, base_depth, timestamp=0.0):
        """intensity: float32 0..255 (CV_32FC1), depth: float32 metres with NaN = invalid (asserts at rgbd_image.cpp:350,356)."""
        I = np.ascontiguousarray(base_intensity)
        Z = np.ascontiguousarray(base_depth)
        assert I.dtype == np.float32 and Z.dtype == np.float32, "intensity and depth must be float32 (CV_32FC1)"
        assert I.shape == (self.height, self.width) and Z.shape == I.shape

        def make(levels):
            ptr = C.c_void_p()
            self.ctx.check(self.ctx._lib.dvo_hip_frame_create_f32(self.ctx.ptr, self.width, self.height, _fp(self.K), _fp(I), _fp(Z),
                                                                   levels, C.byref(ptr)))
            return ptr
        return RgbdImagePyramid(self, make, self.levels, timestamp)

    def create_raw(self, grey_u8, depth_u16, depth_scale=1.0 / 5000.0, timestamp=0.0):
        """Ingest of raw sensor planes (benchmark_slam.cpp:46-93): conversion happens on the device."""
        G = np.ascontiguousarray(grey_u8, dtype=np.uint8)
        D = np.ascontiguousarray(depth_u16, dtype=np.uint16)
        assert G.shape == (self.height, self.width) and D.shape == G.shape

        def make(levels):
            ptr = C.c_void_p()
            self.ctx.check(self.ctx._lib.dvo_hip_frame_create_raw(
                self.ctx.ptr, self.width, self.height, _fp(self.K), G.ctypes.data_as(C.POINTER(C.c_uint8)),
                D.ctypes.data_as(C.POINTER(C.c_uint16)), depth_scale, levels, C.byref(ptr)))
            return ptr
        return RgbdImagePyramid(self, make, self.levels, timestamp)

    def create_raw_device(self, grey_dev_ptr, depth_dev_ptr, depth_scale=1.0 / 5000.0, timestamp=0.0):
        """Raw planes already resident in HBM (device pointers, e.g. torch tensors' data_ptr())."""
        def make(levels):
            ptr = C.c_void_p()
            self.ctx.check(self.ctx._lib.dvo_hip_frame_create_raw_device(
                self.ctx.ptr, self.width, self.height, _fp(self.K), C.c_void_p(grey_dev_ptr), C.c_void_p(depth_dev_ptr),
                depth_scale, levels, C.byref(ptr)))
            return ptr
        return RgbdImagePyramid(self, make, self.levels, timestamp)


_ROLES = {"current": 0, "reference": 1}


class FrameSet:
    """A fixed list of pyramids with its ctypes handle array built once: a streaming caller that re-ingests and aligns the
    same frame objects batch after batch does not rebuild 128-element pointer arrays on every call (the batch entry points
    accept a FrameSet wherever they accept a list of pyramids)."""

    def __init__(self, pyramids):
        self.pyramids = list(pyramids)
        self.n = len(self.pyramids)
        self.ctx = self.pyramids[0].ctx
        self.handles = (C.c_void_p * self.n)(*[p.ptr for p in self.pyramids])

    def __len__(self):
        return self.n

    def __iter__(self):
        return iter(self.pyramids)

    def __getitem__(self, i):
        return self.pyramids[i]


def _handles(frames):
    if isinstance(frames, FrameSet):
        return frames.handles
    return (C.c_void_p * len(frames))(*[p.ptr for p in frames])


def device_pointer_array(ptrs):
    """ctypes array of device addresses (built once by callers that stream from fixed buffers)"""
    return (C.c_void_p * len(ptrs))(*[C.c_void_p(int(x)) for x in ptrs])


def _pointer_array(ptrs):
    return ptrs if isinstance(ptrs, C.Array) else device_pointer_array(ptrs)


def update_raw_device_batch(pyramids, grey_dev_ptrs, depth_dev_ptrs, depth_scale=1.0 / 5000.0, role=None, config=None):
    """Re-ingest raw planes (device pointers) into n existing pyramids of one camera, batched.  With role ("current" /
    "reference") and config: ingest and prepare_roles_batch in one pass over the raw planes (dvo_hip_frames_update_raw_device_as)."""
    n = len(pyramids)
    ctx = pyramids[0].ctx
    fr, g, z = _handles(pyramids), _pointer_array(grey_dev_ptrs), _pointer_array(depth_dev_ptrs)
    if role is None:
        ctx.check(ctx._lib.dvo_hip_frames_update_raw_device(ctx.ptr, n, fr, g, z, depth_scale))
    else:
        ccfg = config if isinstance(config, _lib.Config) else config.to_c()
        ctx.check(ctx._lib.dvo_hip_frames_update_raw_device_as(ctx.ptr, n, fr, g, z, depth_scale, _ROLES[role], C.byref(ccfg)))


class PinnedRawPlanes:
    """Page-locked host memory for the raw planes of n frames (dvo_hip_host_alloc), laid out per frame as [u16 depth][u8 grey]
    so that a frame moves to the device in one transfer.  `depth[i]` / `grey[i]` are numpy views a decoder writes into."""

    def __init__(self, ctx, n, width, height):
        self.ctx, self.n, self.npx = ctx, n, width * height
        stride = (self.npx * 3 + 1) & ~1                     # frames that follow each other at this stride move in one transfer
        ptr = C.c_void_p()
        ctx.check(ctx._lib.dvo_hip_host_alloc(ctx.ptr, n * stride, C.byref(ptr)))
        self.ptr = ptr
        raw = (C.c_uint8 * (n * stride)).from_address(ptr.value)
        block = np.frombuffer(raw, dtype=np.uint8).reshape(n, stride)
        self.depth = [block[i, :self.npx * 2].view(np.uint16).reshape(height, width) for i in range(n)]
        self.grey = [block[i, self.npx * 2:self.npx * 3].reshape(height, width) for i in range(n)]

    def close(self):
        if self.ptr is not None:
            self.depth = self.grey = None
            self.ctx._lib.dvo_hip_host_free(self.ctx.ptr, self.ptr)
            self.ptr = None


def update_raw_host_batch(pyramids, grey_host, depth_host, depth_scale=1.0 / 5000.0, role=None, config=None):
    """Re-ingest raw planes from HOST arrays (uint8 / uint16, C-contiguous) into n existing pyramids: asynchronous DMA on the
    context's upload stream, then the batched build (dvo_hip_frames_update_raw).  The arrays must stay unchanged until
    upload_wait() or until a match on these pyramids has returned."""
    n = len(pyramids)
    ctx = pyramids[0].ctx
    vp = C.c_void_p
    fr = _handles(pyramids)
    if isinstance(grey_host, C.Array):          # addresses prepared once by the caller (device_pointer_array of host addresses)
        g, z = grey_host, depth_host
    else:
        for a, b in zip(grey_host, depth_host):
            assert a.dtype == np.uint8 and b.dtype == np.uint16 and a.flags.c_contiguous and b.flags.c_contiguous
        g = (vp * n)(*[vp(a.ctypes.data) for a in grey_host])
        z = (vp * n)(*[vp(a.ctypes.data) for a in depth_host])
    if role is None:
        ctx.check(ctx._lib.dvo_hip_frames_update_raw(ctx.ptr, n, fr, g, z, depth_scale))
    else:
        ccfg = config if isinstance(config, _lib.Config) else config.to_c()
        ctx.check(ctx._lib.dvo_hip_frames_update_raw_as(ctx.ptr, n, fr, g, z, depth_scale, _ROLES[role], C.byref(ccfg)))


def upload_wait(ctx):
    ctx.check(ctx._lib.dvo_hip_upload_wait(ctx.ptr))


def prepare_roles_batch(pyramids, role, config):
    """Build the role planes of n pyramids ahead of time and asynchronously (dvo_hip_frames_prepare): role "current" = sampling
    planes, "reference" = point selection for config's thresholds.  Together with update_raw_device_batch this runs on the
    context's build stream, concurrently with a match started afterwards on other frames."""
    n = len(pyramids)
    ctx = pyramids[0].ctx
    fr = _handles(pyramids)
    ccfg = config.to_c()
    ctx.check(ctx._lib.dvo_hip_frames_prepare(ctx.ptr, n, fr, _ROLES[role], C.byref(ccfg)))


class PointSelection:
    """Reference-side selection cache (point_selection.h:69-99).  The selected list itself never leaves the GPU."""

    def __init__(self, pyramid=None, intensity_threshold=0.0, depth_threshold=0.0):
        self.pyramid = pyramid
        self.intensity_threshold, self.depth_threshold = intensity_threshold, depth_threshold

    def setRgbdImagePyramid(self, pyramid):
        self.pyramid = pyramid

    recycle = setRgbdImagePyramid

    def getRgbdImagePyramid(self):
        assert self.pyramid is not None
        return self.pyramid

    def getMaximumNumberOfPoints(self, level):     # point_selection.cpp:68-71
        c = self.pyramid.camera
        return int(c.width * c.height * 0.25 ** level)

    def select(self, level, want_mask=False):
        """Returns the number of selected points (and the uint8 mask if asked)."""
        p = self.pyramid
        p.build(level + 1)
        n = C.c_int()
        mask = None
        mp = None
        if want_mask:
            img = p.level(level)
            mask = np.zeros((img.height, img.width), np.uint8)
            mp = mask.ctypes.data_as(C.POINTER(C.c_uint8))
        p.ctx.check(p.ctx._lib.dvo_hip_frame_select(p.ctx.ptr, p.ptr, level, self.intensity_threshold, self.depth_threshold,
                                                   C.byref(n), mp))
        return (n.value, mask) if want_mask else n.value


_RESULT_DTYPE = np.dtype([("transformation", np.float64, (16,)), ("information", np.float64, (36,)), ("loglik", np.float64),
                          ("n_levels", np.int32), ("n_iterations_total", np.int32),
                          ("entropy", np.float64), ("condition_number", np.float64), ("constraint_ratio", np.float64),
                          ("constraint_ratio_accepted", np.float64)])
assert _RESULT_DTYPE.itemsize == C.sizeof(_lib.Result)


def _unpack_stats(res, levels, iters):
    out = []
    for li in range(res.n_levels):
        L = levels[li]
        ls = LevelStats(L.id, L.max_valid_pixels, L.valid_pixels, L.termination)
        for k in range(L.n_iterations):
            s = iters[L.first_iteration_index + k]
            ls.Iterations.append(IterationStats(
                s.id, s.valid_constraints, s.tdist_loglik, np.array(s.tdist_mean), np.array(s.tdist_precision).reshape(2, 2),
                s.prior_loglik, np.array(s.increment), np.array(s.information).reshape(6, 6)))
        out.append(ls)
    return out


class DenseTracker:
    """dvo::DenseTracker (dense_tracking.h:142-162).  Not re-entrant, like the reference: one per thread."""
    _default_config = Config()

    @staticmethod
    def getDefaultConfig():
        return DenseTracker._default_config

    def __init__(self, config=None, ctx=None):
        self.ctx = ctx or default_context()
        self.reference_selection_ = PointSelection()
        self.configure(config or DenseTracker.getDefaultConfig())

    def configure(self, config):
        assert config.IsSane()                                   # dense_tracking.cpp:74
        self.cfg = Config(**config.__dict__)
        self.reference_selection_.intensity_threshold = config.IntensityDerivativeThreshold
        self.reference_selection_.depth_threshold = config.DepthDerivativeThreshold

    def configuration(self):
        return self.cfg

    def match(self, reference, current, result_or_transformation, with_stats=True):
        """match(RgbdImagePyramid|PointSelection reference, RgbdImagePyramid current, Result& | 4x4 ndarray (in/out)).
        Always returns True; failure shows as Result.isNaN() / termination criteria."""
        ref_pyr = reference.getRgbdImagePyramid() if isinstance(reference, PointSelection) else reference
        if isinstance(result_or_transformation, Result):
            self.match_batch([ref_pyr], [current], [result_or_transformation], with_stats=with_stats)
            return True
        T = result_or_transformation
        r = Result()
        r.Transformation = np.array(T, dtype=np.float64)
        self.match_batch([ref_pyr], [current], [r], with_stats=False)
        T[...] = r.Transformation
        return True

    def match_batch(self, references, currents, results, with_stats=False):
        """n independent alignments in one batched launch sequence (keyframe_graph.cpp:576-593 shape)."""
        n = len(references)
        assert len(currents) == n and len(results) == n
        cfg = self.cfg
        for r, c in zip(references, currents):
            r.build(cfg.getNumLevels())                           # dense_tracking.cpp:125, 133
            c.build(cfg.getNumLevels())
        cres = (_lib.Result * n)()
        for i, r in enumerate(results):
            if cfg.UseInitialEstimate:
                assert not r.isNaN(), "Provided initialization is NaN!"   # dense_tracking.cpp:139
            else:
                r.setIdentity()
            for k, v in enumerate(np.asarray(r.Transformation, dtype=np.float64).reshape(-1)):
                cres[i].transformation[k] = v
        vp = C.c_void_p
        refs = (vp * n)(*[p.ptr for p in references])
        curs = (vp * n)(*[p.ptr for p in currents])
        ccfg = cfg.to_c()
        nl = cfg.FirstLevel - cfg.LastLevel + 1
        cap_it = nl * cfg.MaxIterationsPerLevel
        if with_stats:
            levels = (_lib.LevelStats * (n * nl))()
            iters = (_lib.IterationStats * (n * cap_it))()
            rc = self.ctx._lib.dvo_hip_match_batch(self.ctx.ptr, n, refs, curs, C.byref(ccfg), cres, levels, nl, iters, cap_it)
        else:
            rc = self.ctx._lib.dvo_hip_match_batch(self.ctx.ptr, n, refs, curs, C.byref(ccfg), cres, None, 0, None, 0)
        self.ctx.check(rc)
        for i, r in enumerate(results):
            r.Transformation = np.array(cres[i].transformation).reshape(4, 4)
            r.Information = np.array(cres[i].information).reshape(6, 6)
            r.LogLikelihood = cres[i].loglik
            # keyframe-selection statistics computed on the device (extension over the reference's Result)
            r.Entropy, r.ConditionNumber = cres[i].entropy, cres[i].condition_number
            r.ConstraintRatio, r.ConstraintRatioAccepted = cres[i].constraint_ratio, cres[i].constraint_ratio_accepted
            if with_stats:   # appended, not cleared
                r.Statistics.Levels.extend(_unpack_stats(cres[i], levels[i * nl:(i + 1) * nl], iters[i * cap_it:(i + 1) * cap_it]))
        return True

    def match_batch_arrays(self, references, currents, T_init=None):
        """match_batch without per-pair Python objects: returns dict(T [n,4,4], information [n,6,6], loglik [n],
        n_iterations [n]).  T_init: optional [n,4,4] initial guesses (used when UseInitialEstimate)."""
        n = len(references)
        cfg = self.cfg
        if not (isinstance(references, FrameSet) and isinstance(currents, FrameSet)):   # a FrameSet's pyramids are final
            for r, c in zip(references, currents):
                r.build(cfg.getNumLevels())
                c.build(cfg.getNumLevels())
        cres = (_lib.Result * n)()
        view = np.frombuffer(cres, dtype=_RESULT_DTYPE)
        if cfg.UseInitialEstimate:
            assert T_init is not None and np.isfinite(T_init).all(), "Provided initialization is NaN!"
            view["transformation"][:] = np.asarray(T_init, dtype=np.float64).reshape(n, 16)
        else:
            view["transformation"][:] = np.eye(4).reshape(16)
        refs, curs = _handles(references), _handles(currents)
        ccfg = cfg.to_c()
        self.ctx.check(self.ctx._lib.dvo_hip_match_batch(self.ctx.ptr, n, refs, curs, C.byref(ccfg), cres, None, 0, None, 0))
        return dict(T=view["transformation"].reshape(n, 4, 4).copy(), information=view["information"].reshape(n, 6, 6).copy(),
                    loglik=view["loglik"].copy(), n_iterations=view["n_iterations_total"].copy(), entropy=view["entropy"].copy(),
                    condition_number=view["condition_number"].copy(), constraint_ratio=view["constraint_ratio"].copy(),
                    constraint_ratio_accepted=view["constraint_ratio_accepted"].copy())

    def level_iteration(self, reference, current, level, T34, P_prev=None, first=True, want_residuals=False):
        """One Gauss-Newton linearisation at a fixed estimate (parity entry point, dvo_hip_level_iteration)."""
        T34 = np.ascontiguousarray(np.asarray(T34, dtype=np.float32).reshape(-1)[:12])
        Pp = np.zeros(4, np.float32) if P_prev is None else np.ascontiguousarray(np.asarray(P_prev, np.float32).reshape(-1))
        out = _lib.IterationOut()
        img = current.level(level)
        res = np.empty((img.height, img.width, 2), np.float32) if want_residuals else None
        self.ctx.check(self.ctx._lib.dvo_hip_level_iteration(
            self.ctx.ptr, reference.ptr, current.ptr, level, self.cfg.IntensityDerivativeThreshold, self.cfg.DepthDerivativeThreshold,
            _fp(T34), _fp(Pp), int(first), C.byref(out), _fp(res) if want_residuals else None))
        d = dict(n=out.n, n_selected=out.n_selected, cov=np.array(out.scale_cov), P=np.array(out.precision).reshape(2, 2),
                 neg_ll=out.neg_loglik, A=np.array(out.A).reshape(6, 6), b=np.array(out.b))
        if want_residuals:
            d["residuals"] = res
        return d

    def time_residual_kernel(self, references, currents, level, reps=20, warm_iterations=3):
        """Average duration (ms) of one launch of the fused residual/Jacobian/reduce/log-likelihood kernel (HIP events), after
        `warm_iterations` Gauss-Newton steps on that level (converged transform, t-distribution weights on; 0 = identity)."""
        n = len(references)
        vp = C.c_void_p
        refs = (vp * n)(*[p.ptr for p in references])
        curs = (vp * n)(*[p.ptr for p in currents])
        ms = C.c_float()
        self.ctx.check(self.ctx._lib.dvo_hip_time_residual_kernel(self.ctx.ptr, n, refs, curs, level, warm_iterations, reps, C.byref(ms)))
        return ms.value

    def time_stream_mix(self, references, currents, level, reps=20, with_write=False):
        """Average duration (ms) of a kernel that only streams the planes the level's sweep reads (window sweep: 16 B per pixel, gathering
        sweep: 32 B; with_write: + 8 B written)."""
        n = len(references)
        vp = C.c_void_p
        refs = (vp * n)(*[p.ptr for p in references])
        curs = (vp * n)(*[p.ptr for p in currents])
        ms = C.c_float()
        self.ctx.check(self.ctx._lib.dvo_hip_time_stream_mix(self.ctx.ptr, n, refs, curs, level, int(with_write), reps, C.byref(ms)))
        return ms.value
