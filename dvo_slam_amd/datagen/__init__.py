"""Deterministic synthetic RGB-D pairs -- a data tool shared by tests, bench and smoke.

Not part of the alignment path: it only produces the raw sensor planes (grey u8, depth u16 at 5000 counts/m, 0 = hole)
that both the GPU path and the CPU oracle then ingest, so both sides read identical bytes.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libdvo_synth.so")
_SRC = os.path.join(_HERE, "synth.cpp")

FR1_K = np.array([517.3, 516.5, 318.6, 255.3], dtype=np.float32)   # dvo_benchmark/src/benchmark_slam.cpp:384


def build(force=False):
    if force or not os.path.exists(_LIB) or os.path.getmtime(_SRC) > os.path.getmtime(_LIB):
        subprocess.check_call(["g++", "-O2", "-fPIC", "-std=c++17", "-pthread", "-shared", "-o", _LIB, _SRC])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        fp, dp = C.POINTER(C.c_float), C.POINTER(C.c_double)
        u8, u16 = C.POINTER(C.c_uint8), C.POINTER(C.c_uint16)
        L.dvo_synth_pair.argtypes = [C.c_uint64, C.c_int, C.c_int, fp, u8, u16, u8, u16, dp]
        L.dvo_synth_batch.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int, fp, u8, u16, u8, u16, dp, C.c_int]
        L.dvo_synth_sequence.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int, fp, u8, u16, dp, C.c_int]
        L.dvo_synth_sequence_noisy.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int, fp, u8, u16, dp, C.c_double, C.c_double, C.c_double, C.c_int]
        _lib = L
    return _lib


def synth_pair(seed, w=640, h=480, K=None):
    """-> dict(grey_ref u8, depth_ref u16, grey_cur u8, depth_cur u16, xi_true(6), K).  xi_true is the twist (v, omega)
    of the transform match() should return (current -> reference)."""
    b = synth_batch(seed, 1, w, h, K, nthreads=1)
    return dict(grey_ref=b["grey_ref"][0], depth_ref=b["depth_ref"][0], grey_cur=b["grey_cur"][0], depth_cur=b["depth_cur"][0],
                xi_true=b["xi_true"][0], K=b["K"])


def synth_batch(seed0, n, w=640, h=480, K=None, nthreads=None):
    """n pairs with seeds seed0..seed0+n-1 as stacked arrays [n, h, w]."""
    K = np.ascontiguousarray(FR1_K * (w / 640.0) if K is None else K, dtype=np.float32)
    gr = np.empty((n, h, w), np.uint8); dr = np.empty((n, h, w), np.uint16)
    gc = np.empty((n, h, w), np.uint8); dc = np.empty((n, h, w), np.uint16)
    xi = np.zeros((n, 6))
    u8, u16 = C.POINTER(C.c_uint8), C.POINTER(C.c_uint16)
    lib().dvo_synth_batch(seed0, n, w, h, K.ctypes.data_as(C.POINTER(C.c_float)), gr.ctypes.data_as(u8), dr.ctypes.data_as(u16),
                          gc.ctypes.data_as(u8), dc.ctypes.data_as(u16), xi.ctypes.data_as(C.POINTER(C.c_double)),
                          nthreads or min(8, os.cpu_count() or 1))
    return dict(grey_ref=gr, depth_ref=dr, grey_cur=gc, depth_cur=dc, xi_true=xi, K=K)


def synth_sequence(seed, n, w=640, h=480, K=None, nthreads=None, depth_noise=0.0, grey_noise=1.5, exposure=0.0):
    """A camera sweep over one scene: dict(grey [n,h,w] u8, depth [n,h,w] u16, poses [n,4,4] camera->world (frame 0 =
    identity), K).  match(frame k-1, frame k) should return poses[k-1]^-1 poses[k].  depth_noise: multiples of the Kinect depth
    uncertainty the reference assumes (dense_tracking_impl.cpp:122-128), Gaussian; grey_noise: half-width of the uniform intensity
    noise in grey levels; exposure: amplitude of a smooth per-frame gain (1 +- exposure) and bias (+- 100 exposure grey levels) drift."""
    K = np.ascontiguousarray(FR1_K * (w / 640.0) if K is None else K, dtype=np.float32)
    grey = np.empty((n, h, w), np.uint8)
    depth = np.empty((n, h, w), np.uint16)
    poses = np.zeros((n, 4, 4))
    lib().dvo_synth_sequence_noisy(seed, n, w, h, K.ctypes.data_as(C.POINTER(C.c_float)), grey.ctypes.data_as(C.POINTER(C.c_uint8)),
                                   depth.ctypes.data_as(C.POINTER(C.c_uint16)), poses.ctypes.data_as(C.POINTER(C.c_double)),
                                   float(depth_noise), float(grey_noise), float(exposure), nthreads or min(16, os.cpu_count() or 1))
    return dict(grey=grey, depth=depth, poses=poses, K=K)
