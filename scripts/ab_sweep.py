#!/usr/bin/env python
"""A/B of two or more builds of libdvo_hip.so on ONE box: the finest-level sweep (and level 1) at 1024 pairs, converged transform,
weights on, timed in alternation (ABAB...), one process per measurement (DVO_HIP_LIBRARY selects the build).
usage: ab_sweep.py <lib A>[@rows per wave][#variant] <lib B> ... [--rounds N] [--variant V]       (child mode: ab_sweep.py --child <variant>)"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "--child" in sys.argv:
    sys.path.insert(0, HERE)
    import dvo_slam_amd as d
    from dvo_slam_amd import datagen
    v = int(sys.argv[sys.argv.index("--child") + 1])
    n = 1024
    b = datagen.synth_batch(0, 128, 640, 480)
    ctx = d.Context(0)
    ctx.set_option("variant", v)
    ctx.set_option("resident", 0)
    if os.environ.get("AB_RPW"):
        ctx.set_option("rows_per_wave", int(os.environ["AB_RPW"]))     # (levels of the gathering sweep only; the window sweep's tile is fixed)
    cam = d.RgbdCameraPyramid(640, 480, b["K"], ctx)
    cam.build(4)
    refs = [cam.create_raw(b["grey_ref"][i % 128], b["depth_ref"][i % 128]) for i in range(n)]
    curs = [cam.create_raw(b["grey_cur"][i % 128], b["depth_cur"][i % 128]) for i in range(n)]
    trk = d.DenseTracker(d.Config(FirstLevel=3, LastLevel=0), ctx)
    out = []
    for level in (0, 1, 2, 3):
        out.append(min(trk.time_residual_kernel(refs, curs, level, reps=10, warm_iterations=3) for _ in range(4)))
    print("%.4f %.4f %.4f %.4f" % tuple(out))
    sys.exit(0)

args = [a for a in sys.argv[1:]]
rounds, variant = 3, 7
if "--rounds" in args:
    k = args.index("--rounds"); rounds = int(args[k + 1]); del args[k:k + 2]
if "--variant" in args:
    k = args.index("--variant"); variant = int(args[k + 1]); del args[k:k + 2]
libs = args
res = {lib: [] for lib in libs}
for r in range(rounds):
    for lib in libs:
        spec, _, var = lib.partition("#")                       # "<lib>[@<rows per wave>][#<variant>]"
        path, _, rpw = spec.partition("@")
        env = dict(os.environ, DVO_HIP_LIBRARY=os.path.join(HERE, path))
        if rpw:
            env["AB_RPW"] = rpw
        o = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", var or str(variant)], env=env, capture_output=True, text=True)
        if o.returncode != 0:
            print(lib, "failed:", o.stderr[-400:])
            continue
        res[lib].append([float(x) for x in o.stdout.split()[-4:]])
for lib in libs:
    l0 = sorted(x[0] for x in res[lib]); l1 = sorted(x[1] for x in res[lib])
    if l0:
        print("%-40s level 0: %s ms (min %.4f -> %.3f of 8 TB/s at 40 B/px)   level 1: %s (min %.4f)   levels 2, 3: min %.4f, %.4f"
              % (lib, " ".join("%.4f" % x for x in l0), l0[0], 40.0 * 307200 * 1024 / l0[0] / 1e6 / 8000.0, " ".join("%.4f" % x for x in l1), l1[0],
                 min(x[2] for x in res[lib]), min(x[3] for x in res[lib])), flush=True)
