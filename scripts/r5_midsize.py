#!/usr/bin/env python
"""Round 5: the streaming step of a mid-size batch (what each of 8 GPUs runs in BASELINE config 4: 128 pairs) under library options,
configurations alternated in one process on one box.
usage: r5_midsize.py <pairs> <reps> "key=v,key=v" "key=v" ...      (an empty string = the defaults; the first is the baseline)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                         # noqa: E402
import dvo_slam_amd as d            # noqa: E402
from dvo_slam_amd import datagen    # noqa: E402
from dvo_slam_amd.stream import StreamPipeline   # noqa: E402

W, H = 640, 480
B = int(sys.argv[1])
reps = int(sys.argv[2])
configs = sys.argv[3:] or [""]
DEFAULTS = {"build_workgroups": 256}
dev = torch.device("cuda", 0)
b = datagen.synth_batch(0, B, W, H, nthreads=min(32, os.cpu_count() or 8))
grey = torch.from_numpy(np.concatenate([b["grey_ref"], b["grey_cur"]])).to(dev)
depth = torch.from_numpy(np.concatenate([b["depth_ref"], b["depth_cur"]]).view(np.int16)).to(dev)
torch.cuda.synchronize()
gp = [grey[i].data_ptr() for i in range(2 * B)]
zp = [depth[i].data_ptr() for i in range(2 * B)]
ctx = d.Context(0)
cam = d.RgbdCameraPyramid(W, H, b["K"], ctx)
cam.build(4)
sets = [[cam.create_raw_device(gp[i], zp[i]) for i in range(2 * B)] for _ in range(2)]
cfg = d.Config(FirstLevel=3, LastLevel=0, MaxIterationsPerLevel=100, Precision=5e-7, Mu=0.0)
pipe = StreamPipeline(ctx, cfg, [d.FrameSet(fs[:B]) for fs in sets], [d.FrameSet(fs[B:]) for fs in sets], gp[:B], zp[:B], gp[B:], zp[B:])
touched = set()


def apply(spec):
    opts = dict(DEFAULTS)
    for key in touched:
        opts.setdefault(key, {"resident": -1, "table_cache": 1, "compact_residuals": 1, "rendezvous": 1}.get(key, 0))
    for kv in filter(None, spec.split(",")):
        k, _, v = kv.partition("=")
        opts[k] = int(v)
        touched.add(k)
    for k, v in opts.items():
        ctx.set_option(k, v)


KEYS = ("host_batches", "host_ns_prepare", "host_ns_enqueue", "host_ns_wait", "host_ns_finish", "table_uploads_skipped")
host = {}


def run(spec, n):
    apply(spec)
    pipe.step(now=None, nxt=0)
    for j in range(3):
        pipe.step(now=j % 2, nxt=(j + 1) % 2)
    torch.cuda.synchronize()
    c0 = [ctx.counter(k) for k in KEYS]
    t0 = time.perf_counter()
    for j in range(3, 3 + n):
        res = pipe.step(now=j % 2, nxt=(j + 1) % 2)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    c1 = [ctx.counter(k) for k in KEYS]
    nb = max(c1[0] - c0[0], 1)
    host[spec] = "host us per match call: prepare %.0f enqueue %.0f wait %.0f finish %.0f; outside the match %.0f; uploads skipped per step %.1f" % (
        (c1[1] - c0[1]) / nb / 1e3, (c1[2] - c0[2]) / nb / 1e3, (c1[3] - c0[3]) / nb / 1e3, (c1[4] - c0[4]) / nb / 1e3,
        ms * 1e3 - sum(c1[k] - c0[k] for k in (1, 2, 3, 4)) / nb / 1e3, (c1[5] - c0[5]) / n)
    T = res["transformation"].reshape(B, 4, 4).copy()
    return ms, T


base_T = None
table = {c: [] for c in configs}
for rnd in range(3):
    for c in configs:
        ms, T = run(c, reps)
        table[c].append(ms)
        if base_T is None:
            base_T = T
print("%d pairs per step, %d steps per measurement, three rounds alternated:" % (B, reps))
for c in configs:
    ms, T = run(c, 2)
    print("  %-48s %s   median %.3f ms   max |dT| vs first %.1e" % (c or "(defaults)", " ".join("%.3f" % x for x in table[c]), float(np.median(table[c])),
                                                                     float(np.abs(T - base_T).max())))
    print("      " + host[c])
