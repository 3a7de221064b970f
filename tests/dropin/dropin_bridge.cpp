// tests/dropin/dropin_bridge.cpp -- TEST INFRASTRUCTURE (see the Makefile next to it).  C entry points that drive the reference's
// own tracking front end / proposal validator / DenseTracker call pattern -- compiled here against include/dvo/ (the MI355X
// engine's facade) -- with the very source text that oracle/ref_bridge.cpp compiles against the reference's dvo_core.
#include <atomic>
#include <chrono>
#include <cstring>
#include <thread>
#include <vector>

#include <dvo/dense_tracking.h>

#include <dvo_slam/constraints/constraint_proposal_validator.h>
#include <dvo_slam/keyframe_tracker.h>

#include "dvo_oracle.h"   // result / statistics records (layout shared with the oracle's C API and with include/dvo_hip.h)

using namespace dvo::core;

extern int g_ref_completed_local_maps;   // oracle/ref_graph_stub.cpp

extern "C" {

#define DVO_BRIDGE(name) dropin_##name
#include "ref_public_api.inc"
#undef DVO_BRIDGE

// which engine is behind the facade (the test asserts it is the GPU library, not a second copy of the reference)
const char* dropin_engine(void) { return dvo_hip_version(); }

// host mirrors of the facade's RgbdImage: level `level` planes {I, Z, Idx, Idy, Zdx, Zdy} through the PUBLIC FIELDS after the calls
// that fill them in the reference (calculateDerivatives), with mirrors enabled
int dropin_level_fields(int w, int h, const float K[4], const float* intensity, const float* depth, int level, float* planes /* [6][h_l][w_l] */) {
  IntrinsicMatrix intrinsics = IntrinsicMatrix::create(K[0], K[1], K[2], K[3]);
  RgbdCameraPyramid camera(size_t(w), size_t(h), intrinsics);
  camera.build(size_t(level) + 1);
  cv::Mat mi(h, w, CV_32FC1), mz(h, w, CV_32FC1);
  std::memcpy(mi.data, intensity, size_t(w) * h * sizeof(float));
  std::memcpy(mz.data, depth, size_t(w) * h * sizeof(float));
  RgbdImagePyramidPtr pyramid = camera.create(mi, mz);
  const bool before = RgbdImage::hostMirrors();
  RgbdImage::hostMirrors(true);
  RgbdImage& img = pyramid->level(size_t(level));
  img.buildPointCloud();
  img.calculateDerivatives();
  RgbdImage::hostMirrors(before);
  const cv::Mat* fields[6] = {&img.intensity, &img.depth, &img.intensity_dx, &img.intensity_dy, &img.depth_dx, &img.depth_dy};
  const size_t n = size_t(img.width) * img.height;
  for (int k = 0; k < 6; ++k) {
    if (fields[k]->total() != n) return -1 - k;
    std::memcpy(planes + size_t(k) * n, fields[k]->data, n * sizeof(float));
  }
  return int(img.pointcloud.cols());
}

}  // extern "C"
