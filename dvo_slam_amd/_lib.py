"""ctypes binding of libdvo_hip.so (the C-ABI declared in include/dvo_hip.h).

There is no CPU fallback: if the shared library is missing or no gfx950 device is usable, every
entry point raises.  The library is built in-tree by `dvo_slam_amd.build()` (hipcc, gfx950).
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DVO_HIP_LIBRARY") or os.path.join(_HERE, "lib", "libdvo_hip.so")   # override: A/B measurements of two builds
CSRC = os.path.join(_HERE, "csrc")

MAX_LEVELS = 8

OK, ERR_NO_DEVICE, ERR_INVALID, ERR_HIP, ERR_CAPACITY = 0, -1, -2, -3, -4


class DvoHipError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libdvo_hip error %d: %s" % (code, message))
        self.code = code


class Config(C.Structure):
    """dvo_hip_config <-> dvo::DenseTracker::Config (dvo_core/include/dvo/dense_tracking.h:42-69)."""
    _fields_ = [
        ("first_level", C.c_int32), ("last_level", C.c_int32),
        ("max_iterations_per_level", C.c_int32), ("use_initial_estimate", C.c_int32),
        ("precision", C.c_double), ("mu", C.c_double),
        ("intensity_derivative_threshold", C.c_float), ("depth_derivative_threshold", C.c_float),
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
                ("n_levels", C.c_int32), ("n_iterations_total", C.c_int32),
                ("entropy", C.c_double), ("condition_number", C.c_double), ("constraint_ratio", C.c_double),
                ("constraint_ratio_accepted", C.c_double)]


class IterationOut(C.Structure):
    _fields_ = [("n", C.c_int32), ("n_selected", C.c_int32), ("scale_cov", C.c_float * 3), ("precision", C.c_float * 4),
                ("neg_loglik", C.c_double), ("A", C.c_double * 36), ("b", C.c_double * 6), ("sum_w", C.c_double)]


# every symbol include/dvo_hip.h declares (tests check that the library exports all of them)
EXPORTS = [
    "dvo_hip_context_create", "dvo_hip_context_destroy", "dvo_hip_last_error", "dvo_hip_context_stream",
    "dvo_hip_device_count", "dvo_hip_frame_create_f32", "dvo_hip_frame_create_raw", "dvo_hip_frame_create_raw_device",
    "dvo_hip_frame_update_raw_device", "dvo_hip_frames_update_raw_device", "dvo_hip_frames_update_raw", "dvo_hip_frames_update_raw_device_as", "dvo_hip_frames_update_raw_as", "dvo_hip_upload_wait",
    "dvo_hip_host_alloc", "dvo_hip_host_free", "dvo_hip_frames_prepare", "dvo_hip_frame_destroy", "dvo_hip_frame_info", "dvo_hip_frame_download_plane", "dvo_hip_frame_select",
    "dvo_hip_match", "dvo_hip_match_batch", "dvo_hip_level_iteration", "dvo_hip_time_residual_kernel", "dvo_hip_time_stream_mix",
    "dvo_hip_set_option", "dvo_hip_get_counter", "dvo_hip_version",
    "dvo_hip_frames_update_raw_device_as_ex", "dvo_hip_frames_update_raw_as_ex", "dvo_hip_flush_deferred", "dvo_hip_context_device",
    "dvo_hip_comm_get_unique_id", "dvo_hip_comm_create", "dvo_hip_comm_destroy", "dvo_hip_comm_rank", "dvo_hip_comm_size",
    "dvo_hip_comm_last_error", "dvo_hip_gather_records_begin", "dvo_hip_gather_records_end", "dvo_hip_gather_records",
]

ROLE_CURRENT, ROLE_REFERENCE = 0, 1
INGEST_DEFER, INGEST_NO_RAW_COPY = 1, 2
COMM_ID_BYTES = 128


def build(force=False):
    """Compile libdvo_hip.so for gfx950 with the committed Makefile (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-s", "-j4"]
    if force:
        subprocess.check_call(["make", "-C", CSRC, "-s", "clean"])
    subprocess.check_call(cmd)
    # the replay driver (g++ only) links the library just built
    subprocess.check_call(["make", "-C", os.path.join(os.path.dirname(CSRC), "apps"), "-s"])
    return LIB_PATH


_lib = None


def lib():
    """Load libdvo_hip.so. Raises if it has not been built -- never falls back to anything else."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DvoHipError(ERR_NO_DEVICE, "%s not built; run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, fp, ip = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)
    L.dvo_hip_context_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.dvo_hip_context_destroy.argtypes = [vp]
    L.dvo_hip_context_destroy.restype = None
    L.dvo_hip_last_error.argtypes = [vp]
    L.dvo_hip_last_error.restype = C.c_char_p
    L.dvo_hip_context_stream.argtypes = [vp]
    L.dvo_hip_context_stream.restype = vp
    L.dvo_hip_device_count.restype = C.c_int
    L.dvo_hip_frame_create_f32.argtypes = [vp, C.c_int, C.c_int, fp, fp, fp, C.c_int, C.POINTER(vp)]
    L.dvo_hip_frame_create_raw.argtypes = [vp, C.c_int, C.c_int, fp, C.POINTER(C.c_uint8), C.POINTER(C.c_uint16), C.c_float,
                                           C.c_int, C.POINTER(vp)]
    L.dvo_hip_frame_create_raw_device.argtypes = [vp, C.c_int, C.c_int, fp, vp, vp, C.c_float, C.c_int, C.POINTER(vp)]
    L.dvo_hip_frame_update_raw_device.argtypes = [vp, vp, vp, vp, C.c_float]
    L.dvo_hip_frames_update_raw_device.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.c_float]
    L.dvo_hip_frames_update_raw.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.c_float]
    L.dvo_hip_frames_update_raw_device_as.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.c_float, C.c_int, C.POINTER(Config)]
    L.dvo_hip_frames_update_raw_as.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.c_float, C.c_int, C.POINTER(Config)]
    L.dvo_hip_upload_wait.argtypes = [vp]
    L.dvo_hip_host_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.dvo_hip_host_free.argtypes = [vp, vp]
    L.dvo_hip_host_free.restype = None
    L.dvo_hip_frames_prepare.argtypes = [vp, C.c_int, C.POINTER(vp), C.c_int, C.POINTER(Config)]
    L.dvo_hip_frame_destroy.argtypes = [vp, vp]
    L.dvo_hip_frame_destroy.restype = None
    L.dvo_hip_frame_info.argtypes = [vp, C.c_int, ip, ip, fp]
    L.dvo_hip_frame_download_plane.argtypes = [vp, vp, C.c_int, C.c_int, fp]
    L.dvo_hip_frame_select.argtypes = [vp, vp, C.c_int, C.c_float, C.c_float, ip, C.POINTER(C.c_uint8)]
    L.dvo_hip_match.argtypes = [vp, vp, vp, C.POINTER(Config), C.POINTER(Result), C.POINTER(LevelStats), C.c_int,
                                C.POINTER(IterationStats), C.c_int]
    L.dvo_hip_match_batch.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(Config), C.POINTER(Result),
                                      C.POINTER(LevelStats), C.c_int, C.POINTER(IterationStats), C.c_int]
    L.dvo_hip_level_iteration.argtypes = [vp, vp, vp, C.c_int, C.c_float, C.c_float, fp, fp, C.c_int,
                                          C.POINTER(IterationOut), fp]
    L.dvo_hip_time_residual_kernel.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(vp), C.c_int, C.c_int, C.c_int, fp]
    L.dvo_hip_time_stream_mix.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(vp), C.c_int, C.c_int, C.c_int, fp]
    L.dvo_hip_set_option.argtypes = [vp, C.c_char_p, C.c_int]
    L.dvo_hip_get_counter.argtypes = [vp, C.c_char_p, C.POINTER(C.c_longlong)]
    L.dvo_hip_version.restype = C.c_char_p
    L.dvo_hip_frames_update_raw_device_as_ex.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.c_float, C.c_int, C.POINTER(Config), C.c_uint]
    L.dvo_hip_frames_update_raw_as_ex.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.c_float, C.c_int, C.POINTER(Config), C.c_uint]
    L.dvo_hip_flush_deferred.argtypes = [vp]
    L.dvo_hip_context_device.argtypes = [vp]
    L.dvo_hip_comm_get_unique_id.argtypes = [vp]
    L.dvo_hip_comm_create.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(vp)]
    L.dvo_hip_comm_destroy.argtypes = [vp]
    L.dvo_hip_comm_destroy.restype = None
    L.dvo_hip_comm_rank.argtypes = [vp]
    L.dvo_hip_comm_size.argtypes = [vp]
    L.dvo_hip_comm_last_error.argtypes = [vp]
    L.dvo_hip_comm_last_error.restype = C.c_char_p
    L.dvo_hip_gather_records_begin.argtypes = [vp, vp, C.c_size_t, C.c_size_t, ip]
    L.dvo_hip_gather_records_end.argtypes = [vp, C.c_int, vp, C.c_size_t]
    L.dvo_hip_gather_records.argtypes = [vp, vp, C.c_size_t, C.c_size_t, vp, C.c_size_t]
    _lib = L
    return L
