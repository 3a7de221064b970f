// oracle/ref_bridge.cpp -- TEST INFRASTRUCTURE.  C entry points around the REFERENCE's own alignment code.
//
// oracle/_ref/libdvo_ref.so = twelve translation units of dvo_core (oracle/Makefile: REF_UNITS -- the DenseTracker driver, its SSE
// passes, the normal equations, point selection, the RGB-D image model, intrinsics, raw-depth ingest), compiled unmodified
// from /root/reference (never copied) against the stand-in headers in oracle/shim/, plus this file.  It exists to PIN the
// oracle: tests/test_oracle_ref.py feeds both the same arrays and compares.  Nothing but tests may load it.
//
// What is the reference's: every line of control flow and every float operation of DenseTracker::match(), the SSE passes,
// pyramid / derivative / point-cloud / acceleration-structure / selection code.  What the stand-ins supply (none of it is in
// /root/reference: Eigen, OpenCV, Sophus, boost, TBB are external, un-vendored dependencies): containers (cv::Mat,
// fixed-size Eigen matrices, cv::merge), the small fixed-size algebra (K*T, 2x2 inverse / determinant, 1x2 * 2x6 products,
// the b-vector update), SE(3) exp / log / compose and the 6x6 LDL^T solve -- the last two are the oracle's own
// (oracle/se3_oracle.h), so a whole-match comparison isolates the driver's control flow and the float passes.
#include <atomic>
#include <chrono>
#include <cstring>
#include <thread>
#include <vector>

#include <dvo/dense_tracking_impl.h>
#include <dvo/core/surface_pyramid.h>

#include <dvo_slam/constraints/constraint_proposal_validator.h>
#include <dvo_slam/keyframe_tracker.h>

#include "dvo_oracle.h"   // result / statistics records shared with the oracle's C API

using namespace dvo::core;

extern int g_ref_completed_local_maps;   // ref_graph_stub.cpp

namespace {

typedef dvo::DenseTracker::ResidualVectorType ResidualVector;
typedef dvo::DenseTracker::WeightVectorType WeightVector;

Eigen::Matrix2f matrix2(const float* p_rowmajor) {
  Eigen::Matrix2f m;
  m(0, 0) = p_rowmajor[0]; m(0, 1) = p_rowmajor[1]; m(1, 0) = p_rowmajor[2]; m(1, 1) = p_rowmajor[3];
  return m;
}

ResidualVector residual_vector(const float* res, int n) {
  ResidualVector v(static_cast<size_t>(n));
  for (int i = 0; i < n; ++i) { v[size_t(i)](0) = res[2 * i]; v[size_t(i)](1) = res[2 * i + 1]; }
  return v;
}

}  // namespace

extern "C" {

// points: n x 12 floats (point xyz1, then i z idx idy zdx zdy t -); accel: h x w x 8 floats; T: row-major 3x4 (reference ->
// current); weights: 8 + 8 floats (dense_tracking.cpp:215-220).  Returns the number of compacted outputs.
int ref_compute_residuals(int sse, int n, const float* points, const float* accel, int w, int h, const float K[4], const float T[12],
                          const float* reference_weight, const float* current_weight, float* out_points, float* out_residuals) {
  IntrinsicMatrix intrinsics = IntrinsicMatrix::create(K[0], K[1], K[2], K[3]);
  RgbdCamera camera(size_t(w), size_t(h), intrinsics);
  RgbdImage current(camera);
  current.width = size_t(w);
  current.height = size_t(h);
  current.acceleration.create(h, w);
  std::memcpy(current.acceleration.data, accel, size_t(w) * h * 8 * sizeof(float));

  PointWithIntensityAndDepth::VectorType in(static_cast<size_t>(n)), out(static_cast<size_t>(n) + 2);
  for (int i = 0; i < n; ++i) {
    std::memcpy(in[size_t(i)].point.data, points + size_t(i) * 12, 4 * sizeof(float));
    std::memcpy(in[size_t(i)].intensity_and_depth.data, points + size_t(i) * 12 + 4, 8 * sizeof(float));
  }
  ResidualVector residuals(static_cast<size_t>(n) + 2);
  Eigen::Affine3f transform;
  transform.setIdentity();
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j) transform.matrix()(i, j) = T[i * 4 + j];
  Vector8f wref, wcur;
  for (int i = 0; i < 8; ++i) { wref(i) = reference_weight[i]; wcur(i) = current_weight[i]; }

  ComputeResidualsResult result;
  result.first_point_error = out.begin();
  result.first_residual = residuals.begin();
  std::vector<uint8_t> flags(static_cast<size_t>(n) + 2);
  result.first_valid_flag = flags.begin();
  if (sse) computeResidualsSse(in.begin(), in.end(), current, intrinsics, transform, wref, wcur, result);
  else computeResiduals(in.begin(), in.end(), current, intrinsics, transform, wref, wcur, result);
  const int n_out = int(result.last_point_error - result.first_point_error);
  for (int i = 0; i < n_out; ++i) {
    std::memcpy(out_points + size_t(i) * 12, out[size_t(i)].point.data, 4 * sizeof(float));
    std::memcpy(out_points + size_t(i) * 12 + 4, out[size_t(i)].intensity_and_depth.data, 8 * sizeof(float));
    out_residuals[2 * i] = residuals[size_t(i)](0);
    out_residuals[2 * i + 1] = residuals[size_t(i)](1);
  }
  return n_out;
}

// precision: row-major 2x2
void ref_compute_weights(int sse, int n, const float* residuals, const float mean[2], const float precision[4], float* weights) {
  ResidualVector r = residual_vector(residuals, n);
  WeightVector w(static_cast<size_t>(n), 0.0f);
  const Eigen::Vector2f m(mean[0], mean[1]);
  const Eigen::Matrix2f P = matrix2(precision);
  if (sse) computeWeightsSse(r.begin(), r.end(), w.begin(), m, P);
  else computeWeights(r.begin(), r.end(), w.begin(), m, P);
  std::memcpy(weights, w.data(), size_t(n) * sizeof(float));
}

// scale: row-major 2x2 out
void ref_compute_scale(int sse, int n, const float* residuals, const float* weights, const float mean[2], float scale[4]) {
  ResidualVector r = residual_vector(residuals, n);
  WeightVector w(weights, weights + n);
  const Eigen::Vector2f m(mean[0], mean[1]);
  const Eigen::Matrix2f S = sse ? computeScaleSse(r.begin(), r.end(), w.begin(), m) : computeScale(r.begin(), r.end(), w.begin(), m);
  scale[0] = S(0, 0); scale[1] = S(0, 1); scale[2] = S(1, 0); scale[3] = S(1, 1);
}

float ref_loglik(int n, const float* residuals, const float* weights, const float mean[2], const float precision[4]) {
  ResidualVector r = residual_vector(residuals, n);
  WeightVector w(weights, weights + n);
  return computeCompleteDataLogLikelihood(r.begin(), r.end(), w.begin(), Eigen::Vector2f(mean[0], mean[1]), matrix2(precision));
}

// J: n x 12 floats (row-major 2x6 per point), alpha: row-major 2x2; A: row-major 6x6 out (OptimizedSelfAdjointMatrix6x6f)
void ref_rank_update_2x6(int n, const float* J, const float alpha[4], float A[36]) {
  OptimizedSelfAdjointMatrix6x6f acc;
  acc.setZero();
  const Eigen::Matrix2f a = matrix2(alpha);
  for (int i = 0; i < n; ++i) {
    Eigen::Matrix<float, 2, 6> j;
    for (int r = 0; r < 2; ++r)
      for (int c = 0; c < 6; ++c) j(r, c) = J[size_t(i) * 12 + r * 6 + c];
    acc.rankUpdate(j, a);
  }
  Eigen::Matrix<float, 6, 6> m;
  acc.toEigen(m);
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c < 6; ++c) A[r * 6 + c] = m(r, c);
}

// SurfacePyramid::convertRawDepthImage[Sse] (surface_pyramid.cpp:45-105): u16 * scale, 0 -> NaN
void ref_convert_raw_depth(int sse, const uint16_t* raw, int w, int h, float scale, float* out) {
  cv::Mat in(h, w, CV_16UC1), res;
  std::memcpy(in.data, raw, size_t(w) * h * sizeof(uint16_t));
  if (sse) SurfacePyramid::convertRawDepthImageSse(in, res, scale);
  else SurfacePyramid::convertRawDepthImage(in, res, scale);
  std::memcpy(out, res.data, size_t(w) * h * sizeof(float));
}

// IntrinsicMatrix::scale (intrinsic_matrix.cpp:90-93)
void ref_intrinsics_scale(const float K[4], float factor, float out[4]) {
  IntrinsicMatrix m = IntrinsicMatrix::create(K[0], K[1], K[2], K[3]);
  m.scale(factor);
  out[0] = m.fx(); out[1] = m.fy(); out[2] = m.ox(); out[3] = m.oy();
}

// The entry points that use nothing but the PUBLIC class API (DenseTracker::match, the validator, the tracking front end) are shared
// with tests/dropin/ (where the same source is compiled against this engine's facade headers instead of the reference's dvo_core):
#define DVO_BRIDGE(name) ref_##name
#include "ref_public_api.inc"
#undef DVO_BRIDGE

// one level of the reference's image model (rgbd_image.cpp:156-172, 419-543; point_selection.cpp:89-152): the six planes
// [6][h_l][w_l], the selection mask and (optionally) the selected points of pyramid level `level`; returns their number
int ref_level_planes(int w, int h, const float K[4], const float* intensity, const float* depth, int level, float* planes,
                     unsigned char* mask, float K_level[4], float* points /* n x 12 or null */) {
  IntrinsicMatrix intrinsics = IntrinsicMatrix::create(K[0], K[1], K[2], K[3]);
  RgbdCameraPyramid camera(size_t(w), size_t(h), intrinsics);
  camera.build(size_t(level) + 1);
  cv::Mat I(h, w, CV_32FC1), Z(h, w, CV_32FC1);
  std::memcpy(I.data, intensity, size_t(w) * h * sizeof(float));
  std::memcpy(Z.data, depth, size_t(w) * h * sizeof(float));
  RgbdImagePyramidPtr pyramid = camera.create(I, Z);
  pyramid->build(size_t(level) + 1);
  RgbdImage& img = pyramid->level(size_t(level));
  img.buildPointCloud();
  img.buildAccelerationStructure();
  const int lw = int(img.width), lh = int(img.height);
  const cv::Mat* src[6] = {&img.intensity, &img.depth, &img.intensity_dx, &img.intensity_dy, &img.depth_dx, &img.depth_dy};
  for (int p = 0; p < 6; ++p) std::memcpy(planes + size_t(p) * lw * lh, src[p]->data, size_t(lw) * lh * sizeof(float));
  const IntrinsicMatrix& Kl = camera.level(size_t(level)).intrinsics();
  K_level[0] = Kl.fx(); K_level[1] = Kl.fy(); K_level[2] = Kl.ox(); K_level[3] = Kl.oy();
  // selection with the tracker's predicate at thresholds 0 (point_selection.h:49-67); the debug index marks the selected pixels
  ValidPointAndGradientThresholdPredicate predicate;
  PointSelection selection(*pyramid, predicate);
  selection.debug(true);
  PointSelection::PointIterator first, last;
  selection.select(size_t(level), first, last);
  const int n_selected = int(last - first);
  cv::Mat index;
  selection.getDebugIndex(size_t(level), index);
  std::memcpy(mask, index.data, size_t(lw) * lh);
  if (points)
    for (int i = 0; i < n_selected; ++i) {
      std::memcpy(points + size_t(i) * 12, (first + i)->point.data, 4 * sizeof(float));
      std::memcpy(points + size_t(i) * 12 + 4, (first + i)->intensity_and_depth.data, 8 * sizeof(float));
    }
  return n_selected;
}

}  // extern "C"
