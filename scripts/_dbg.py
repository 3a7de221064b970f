import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import dvo_slam_amd as d
from dvo_slam_amd import datagen
from oracle import pyoracle as po
np.set_printoptions(linewidth=200, precision=4)
pair = datagen.synth_pair(14, 128, 96)
T34 = po.se3_exp(np.zeros(6))[:3]
for v in (5, 6):
    ctx = d.Context(0); ctx.set_option("variant", v); ctx.set_option("resident", 0); ctx.set_option("rows_per_wave", 4)
    cam = d.RgbdCameraPyramid(128, 96, pair["K"], ctx); cam.build(1)
    ref, cur = cam.create_raw(pair["grey_ref"], pair["depth_ref"]), cam.create_raw(pair["grey_cur"], pair["depth_cur"])
    trk = d.DenseTracker(d.Config(FirstLevel=0, LastLevel=0), ctx)
    o = trk.level_iteration(ref, cur, 0, T34, first=True)
    print(v, o["n"], o["cov"], "\n", o["A"], "\n", o["b"])
