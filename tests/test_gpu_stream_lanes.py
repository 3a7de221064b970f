"""The streaming loop over several lanes on one GPU (dvo_slam_amd/apps/stream_pipeline.cpp, dvo_stream_lanes_*; dvo_slam_amd/stream.py::
StreamLanes): the pairs of a step dealt to G contexts of the device, a host thread each, that run their shards' steps without waiting
for one another -- the reference's model for independent pairs (the workers of a tbb::parallel_reduce run whole match() calls,
dvo_slam/src/keyframe_graph.cpp:576-593) with the GPU's queues in place of the cores.  Every step re-ingests and aligns every pair; a
pair's result is what the one-lane loop gives it, to the precision of the stopping rule (another batch-size class of the schedule)."""
import numpy as np
import pytest
import torch

import common as cm
import dvo_slam_amd as d
from dvo_slam_amd import _lib, datagen
from dvo_slam_amd.stream import StreamLanes, StreamPipeline

pytestmark = pytest.mark.gpu

W, H, B = 320, 240, 150


@pytest.fixture(scope="module")
def planes():
    b = datagen.synth_batch(900, 10, W, H)
    order = [(3 * i + 1) % 10 for i in range(B)]
    dev = torch.device("cuda", 0)
    grey = torch.from_numpy(np.concatenate([b["grey_ref"][order], b["grey_cur"][order]])).to(dev)
    depth = torch.from_numpy(np.concatenate([b["depth_ref"][order], b["depth_cur"][order]]).view(np.int16)).to(dev)
    torch.cuda.synchronize()
    return dict(b=b, order=order, grey=grey, depth=depth, gp=[grey[i].data_ptr() for i in range(2 * B)], zp=[depth[i].data_ptr() for i in range(2 * B)])


def one_lane(planes, cfg, steps):
    ctx = d.Context(0)
    cam = d.RgbdCameraPyramid(W, H, planes["b"]["K"], ctx)
    cam.build(4)
    gp, zp = planes["gp"], planes["zp"]
    sets = [[cam.create_raw_device(gp[i], zp[i]) for i in range(2 * B)] for _ in range(2)]
    pipe = StreamPipeline(ctx, cfg, [d.FrameSet(fs[:B]) for fs in sets], [d.FrameSet(fs[B:]) for fs in sets], gp[:B], zp[:B], gp[B:], zp[B:])
    pipe.step(now=None, nxt=0)
    out = []
    for k in range(steps):
        out.append(pipe.step(now=k % 2, nxt=(k + 1) % 2)["transformation"].reshape(B, 4, 4).copy())
    return out


@pytest.mark.parametrize("n_lanes", [2, 3])
def test_lanes_give_every_pair_the_one_lane_result(planes, n_lanes):
    cfg = d.Config(FirstLevel=3, LastLevel=0)
    base = one_lane(planes, cfg, 2)
    assert np.array_equal(base[0], base[1])                                  # the same raw planes every step
    gp, zp = planes["gp"], planes["zp"]
    ctxs = [d.Context(0) for _ in range(n_lanes)]
    cams = {}
    for c in ctxs:
        cams[id(c)] = d.RgbdCameraPyramid(W, H, planes["b"]["K"], c)
        cams[id(c)].build(4)

    def make_frames(c, idx):
        return ([cams[id(c)].create_raw_device(gp[i], zp[i]) for i in idx], [cams[id(c)].create_raw_device(gp[B + i], zp[B + i]) for i in idx])
    lanes = StreamLanes(ctxs, cfg, B, make_frames, gp[:B], zp[:B], gp[B:], zp[B:], depth=2)
    got = []
    for k in range(5):                                                        # submit ahead, collect behind: the bench's loop
        lanes.submit()
        if lanes.outstanding >= 2:
            got.append(lanes.collect()["transformation"].reshape(B, 4, 4).copy())
    while lanes.outstanding:
        got.append(lanes.collect()["transformation"].reshape(B, 4, 4).copy())
    assert len(got) == 5
    for T in got:
        assert np.isfinite(T).all()
        assert np.array_equal(T, got[0])                                      # every step: the same planes, the same shards, the same bytes
        assert max(cm.twist_matrix_error(T[i], base[0][i]) for i in range(B)) < 2e-6
    # pair i came from lane i % G: copies of one pair that sit in the same lane have the same bytes
    order = planes["order"]
    for i in range(B):
        for j in range(i + n_lanes, B, n_lanes):
            if order[j] == order[i]:
                assert np.array_equal(got[0][i], got[0][j])
                break
    # the protocol's edges: nothing to collect; more than `depth` steps ahead
    with pytest.raises(_lib.DvoHipError):
        lanes.collect()
    lanes.outstanding = 0
    lanes.submit(); lanes.submit()
    with pytest.raises(_lib.DvoHipError):
        lanes.submit()
    lanes.outstanding = 2
    lanes.collect(); lanes.collect()
    lanes.close()
