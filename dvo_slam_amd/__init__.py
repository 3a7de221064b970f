"""dvo_slam_amd -- MI355X-native dense RGB-D alignment (the DenseTracker::match hot path of tum-vision/dvo_slam).

The product is libdvo_hip.so (hand-written HIP for gfx950 behind the C-ABI in include/dvo_hip.h);
this package is the thin Python mirror of the reference's class API used by the tests and bench.
"""
from ._lib import DvoHipError, build, lib, LIB_PATH  # noqa: F401
from .tracker import (Config, Context, DenseTracker, IterationStats, LevelStats, PointSelection, Result,  # noqa: F401
                      RgbdCameraPyramid, RgbdImage, RgbdImagePyramid, Stats, TERMINATION, default_context,
                      update_raw_device_batch, prepare_roles_batch, update_raw_host_batch, upload_wait, PinnedRawPlanes, FrameSet, device_pointer_array)
