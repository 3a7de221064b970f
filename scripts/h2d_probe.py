#!/usr/bin/env python
"""Raw host-to-device rate of this box for the transfer sizes of the from_host leg: pinned memory from the library's allocator
(dvo_hip_host_alloc = hipHostMalloc) and from torch, idle device."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import dvo_slam_amd as d
ctx = d.default_context()
L = ctx._lib
for mb in (118, 472, 944, 1887):
    n = mb * 1000 * 1000
    p = C.c_void_p()
    assert L.dvo_hip_host_alloc(ctx.ptr, C.c_size_t(n), C.byref(p)) == 0
    host = (C.c_uint8 * n).from_address(p.value)
    a = torch.frombuffer(host, dtype=torch.uint8)
    a[::4096] = 1                                   # touch every page
    dev = torch.empty(n, dtype=torch.uint8, device="cuda")
    for name, src in (("hipHostMalloc", a), ("torch pinned", torch.empty(n, dtype=torch.uint8).pin_memory())):
        src[::4096] = 1
        for _ in range(2):
            dev.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            dev.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print("%5d MB  %-14s %.1f GB/s" % (mb, name, n / dt / 1e9), flush=True)
    L.dvo_hip_host_free(ctx.ptr, p)
