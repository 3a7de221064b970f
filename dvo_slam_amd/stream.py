"""The streaming loop of bench.py as one foreign call per step: dvo_slam_amd/apps/stream_pipeline.cpp (a consumer of the
C-ABI in its own shared object, lib/libdvo_stream.so) re-ingests the next batch in the roles its frames will play and aligns
the current one.  Results land in a buffer allocated once."""
import ctypes as C
import os

import numpy as np

from . import _lib
from .tracker import FrameSet, device_pointer_array, _RESULT_DTYPE

_PIPE_PATH = os.path.join(os.path.dirname(os.path.abspath(_lib.LIB_PATH)), "libdvo_stream.so")
_pipe = None


def _load():
    global _pipe
    if _pipe is None:
        _lib.lib()                                   # libdvo_hip.so first: the pipeline object links against it
        if not os.path.exists(_PIPE_PATH):
            raise _lib.DvoHipError(_lib.ERR_NO_DEVICE, "%s not built; run `python -c 'import __graft_entry__ as g; g.build()'`" % _PIPE_PATH)
        L = C.CDLL(_PIPE_PATH)
        vp = C.c_void_p
        pp = C.POINTER(vp)
        L.dvo_stream_step.argtypes = [vp, C.c_int, pp, pp, pp, pp, pp, pp, C.c_float, pp, pp, C.POINTER(_lib.Config), C.POINTER(_lib.Result)]
        L.dvo_stream_step_host.argtypes = L.dvo_stream_step.argtypes
        L.dvo_stream_pack_records.argtypes = [C.c_int, C.POINTER(_lib.Result), C.POINTER(C.c_double)]
        L.dvo_stream_pack_records.restype = None
        _pipe = L
    return _pipe


class StreamPipeline:
    """ref_sets / cur_sets: lists of FrameSet (one per buffer of the pipeline, n frames each); the raw planes of every batch are
    read from the same device addresses (grey_ref ... depth_cur: n device pointers each)."""

    def __init__(self, ctx, config, ref_sets, cur_sets, grey_ref, depth_ref, grey_cur, depth_cur, depth_scale=1.0 / 5000.0):
        self.ctx, self.n = ctx, len(ref_sets[0])
        self.ref_sets = [s if isinstance(s, FrameSet) else FrameSet(s) for s in ref_sets]
        self.cur_sets = [s if isinstance(s, FrameSet) else FrameSet(s) for s in cur_sets]
        self.ptrs = [device_pointer_array(p) for p in (grey_ref, depth_ref, grey_cur, depth_cur)]
        self.scale = depth_scale
        self.ccfg = config.to_c()
        self.cres = (_lib.Result * self.n)()
        self.results = np.frombuffer(self.cres, dtype=_RESULT_DTYPE)      # view: fields of the last aligned batch
        self.L = _load()
        self._records = np.zeros((self.n, 32), np.float64)

    def records(self):
        """The records of the batch aligned last (dvo_slam_amd/parallel.py layout: twist | information triangle | log-likelihood |
        flag), packed by the pipeline object; the array is reused by the next call."""
        self.L.dvo_stream_pack_records(self.n, self.cres, self._records.ctypes.data_as(C.POINTER(C.c_double)))
        return self._records

    def set_host_planes(self, grey_ref, depth_ref, grey_cur, depth_cur):
        """Host arrays (pinned, uint8 / uint16, C-contiguous, n each) the raw planes of every batch are DMA-ed from by step_host."""
        vp = C.c_void_p
        self.host_ptrs = [(vp * self.n)(*[vp(a.ctypes.data) for a in arrays]) for arrays in (grey_ref, depth_ref, grey_cur, depth_cur)]

    def step_host(self, now=None, nxt=None):
        """step() with the raw planes of `nxt` coming from the host arrays of set_host_planes."""
        nr = self.ref_sets[nxt].handles if nxt is not None else None
        nc = self.cur_sets[nxt].handles if nxt is not None else None
        ar = self.ref_sets[now].handles if now is not None else None
        ac = self.cur_sets[now].handles if now is not None else None
        g_ref, z_ref, g_cur, z_cur = self.host_ptrs
        self.ctx.check(self.L.dvo_stream_step_host(self.ctx.ptr, self.n, nr, nc, g_ref, z_ref, g_cur, z_cur, self.scale, ar, ac,
                                                   C.byref(self.ccfg), self.cres))
        return self.results

    def step(self, now=None, nxt=None):
        """Re-ingest buffer `nxt` (None: skip) and align buffer `now` (None: skip).  Returns the result view."""
        nr = self.ref_sets[nxt].handles if nxt is not None else None
        nc = self.cur_sets[nxt].handles if nxt is not None else None
        ar = self.ref_sets[now].handles if now is not None else None
        ac = self.cur_sets[now].handles if now is not None else None
        g_ref, z_ref, g_cur, z_cur = self.ptrs
        self.ctx.check(self.L.dvo_stream_step(self.ctx.ptr, self.n, nr, nc, g_ref, z_ref, g_cur, z_cur, self.scale, ar, ac,
                                              C.byref(self.ccfg), self.cres))
        return self.results
