"""The streaming loop of bench.py as one foreign call per step: dvo_slam_amd/apps/stream_pipeline.cpp (a consumer of the
C-ABI in its own shared object, lib/libdvo_stream.so) re-ingests the next batch in the roles its frames will play and aligns
the current one.  Results land in a buffer allocated once."""
import ctypes as C
import os

import numpy as np

from . import _lib
from .tracker import FrameSet, device_pointer_array, _RESULT_DTYPE

class _Lane(C.Structure):
    """dvo_stream_lane (stream_pipeline.cpp)"""
    _pp = C.POINTER(C.c_void_p)
    _fields_ = [("ctx", C.c_void_p), ("n", C.c_int), ("refs", _pp * 2), ("curs", _pp * 2),
                ("grey_ref", _pp), ("raw_ref", _pp), ("grey_cur", _pp), ("raw_cur", _pp)]


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
        L.dvo_stream_lanes_create.argtypes = [C.c_int, C.POINTER(_Lane), C.c_float, C.POINTER(_lib.Config), C.c_int]
        L.dvo_stream_lanes_create.restype = vp
        L.dvo_stream_lanes_submit.argtypes = [vp]
        L.dvo_stream_lanes_collect.argtypes = [vp, C.POINTER(_lib.Result)]
        L.dvo_stream_lanes_destroy.argtypes = [vp]
        L.dvo_stream_lanes_destroy.restype = None
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


class StreamLanes:
    """The streaming loop over G lanes on ONE GPU (dvo_stream_lanes_*, stream_pipeline.cpp): the n pairs of a step are dealt to the lanes
    -- lane l takes the pairs l, l + G, ... -- each lane a context of its own on the device and a host thread that runs its shard's
    steps without waiting for the others.  make_frames(ctx, indices) -> (ref frames, cur frames) creates one frame set of a lane in that
    lane's context; grey_ref ... depth_cur: the n device pointers the raw planes of every step are read from."""

    def __init__(self, contexts, config, n, make_frames, grey_ref, depth_ref, grey_cur, depth_cur, depth_scale=1.0 / 5000.0, depth=2):
        self.L = _load()
        self.contexts, self.n, self.G = list(contexts), n, len(contexts)
        self.keep = []                                            # everything the C side holds pointers into
        lanes = (_Lane * self.G)()
        for l, ctx in enumerate(self.contexts):
            idx = list(range(l, n, self.G))
            lanes[l].ctx, lanes[l].n = ctx.ptr, len(idx)
            for k in range(2):
                r, c = make_frames(ctx, idx)
                r, c = (r if isinstance(r, FrameSet) else FrameSet(r)), (c if isinstance(c, FrameSet) else FrameSet(c))
                self.keep += [r, c]
                lanes[l].refs[k], lanes[l].curs[k] = C.cast(r.handles, _Lane._pp), C.cast(c.handles, _Lane._pp)
            for name, src in (("grey_ref", grey_ref), ("raw_ref", depth_ref), ("grey_cur", grey_cur), ("raw_cur", depth_cur)):
                arr = device_pointer_array([src[i] for i in idx])
                self.keep.append(arr)
                setattr(lanes[l], name, C.cast(arr, _Lane._pp))
        self.ccfg = config.to_c()
        self.keep.append(lanes)
        self.ptr = self.L.dvo_stream_lanes_create(self.G, lanes, depth_scale, C.byref(self.ccfg), depth)
        if not self.ptr:
            raise _lib.DvoHipError(_lib.ERR_HIP, "dvo_stream_lanes_create failed: " + "; ".join(c._lib.dvo_hip_last_error(c.ptr).decode() for c in self.contexts))
        self.depth = depth
        self.cres = (_lib.Result * n)()
        self.results = np.frombuffer(self.cres, dtype=_RESULT_DTYPE)
        self._records = np.zeros((n, 32), np.float64)
        self.outstanding = 0

    def submit(self):
        """One more step on every lane (returns at once); at most `depth` steps may await their collect()."""
        rc = self.L.dvo_stream_lanes_submit(self.ptr)
        if rc != 0:
            raise _lib.DvoHipError(rc, "dvo_stream_lanes_submit: %d steps are waiting to be collected" % self.outstanding)
        self.outstanding += 1

    def collect(self):
        """Waits for the oldest submitted step; returns the result view (pair order as dealt: pair i from lane i % G)."""
        rc = self.L.dvo_stream_lanes_collect(self.ptr, self.cres)
        self.outstanding -= 1
        if rc != 0:
            raise _lib.DvoHipError(rc, "a lane's step failed: " + "; ".join(c._lib.dvo_hip_last_error(c.ptr).decode() for c in self.contexts))
        return self.results

    def records(self):
        self.L.dvo_stream_pack_records(self.n, self.cres, self._records.ctypes.data_as(C.POINTER(C.c_double)))
        return self._records

    def close(self):
        if self.ptr:
            self.L.dvo_stream_lanes_destroy(self.ptr)
            self.ptr = None

    def __del__(self):
        self.close()
