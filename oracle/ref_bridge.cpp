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

// DenseTracker::match(RgbdImagePyramid&, RgbdImagePyramid&, Result&) of the reference, dvo_core/src/dense_tracking.cpp:123-376, on
// float planes (intensity 0..255, depth in metres, NaN = hole).  result->transformation is in/out.  cfg->mode is ignored.
int ref_match(int w, int h, const float K[4], const float* ref_intensity, const float* ref_depth, const float* cur_intensity,
              const float* cur_depth, const oracle_config* cfg, oracle_result* result, oracle_level_stats* levels, int cap_levels,
              oracle_iteration_stats* iters, int cap_iters) {
  IntrinsicMatrix intrinsics = IntrinsicMatrix::create(K[0], K[1], K[2], K[3]);
  RgbdCameraPyramid camera(size_t(w), size_t(h), intrinsics);
  dvo::DenseTracker::Config c = dvo::DenseTracker::getDefaultConfig();
  c.FirstLevel = cfg->first_level;
  c.LastLevel = cfg->last_level;
  c.MaxIterationsPerLevel = cfg->max_iterations_per_level;
  c.UseInitialEstimate = cfg->use_initial_estimate != 0;
  c.Precision = cfg->precision;
  c.Mu = cfg->mu;
  c.IntensityDerivativeThreshold = cfg->intensity_derivative_threshold;
  c.DepthDerivativeThreshold = cfg->depth_derivative_threshold;
  camera.build(c.getNumLevels());
  auto image = [&](const float* p) {
    cv::Mat m(h, w, CV_32FC1);
    std::memcpy(m.data, p, size_t(w) * h * sizeof(float));
    return m;
  };
  RgbdImagePyramidPtr reference = camera.create(image(ref_intensity), image(ref_depth));
  RgbdImagePyramidPtr current = camera.create(image(cur_intensity), image(cur_depth));

  dvo::DenseTracker tracker(c);
  dvo::DenseTracker::Result r;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) r.Transformation.matrix()(i, j) = result->transformation[i * 4 + j];
  tracker.match(*reference, *current, r);

  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) result->transformation[i * 4 + j] = r.Transformation.matrix()(i, j);
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) result->information[i * 6 + j] = r.Information(i, j);
  result->loglik = r.LogLikelihood;
  result->n_levels = int(r.Statistics.Levels.size());
  int n_it = 0;
  for (size_t l = 0; l < r.Statistics.Levels.size(); ++l) {
    const dvo::DenseTracker::LevelStats& L = r.Statistics.Levels[l];
    if (int(l) < cap_levels) {
      oracle_level_stats& o = levels[l];
      o.id = int(L.Id);
      o.max_valid_pixels = int(L.MaxValidPixels);
      o.valid_pixels = int(L.ValidPixels);
      o.termination = int(L.TerminationCriterion);
      o.n_iterations = int(L.Iterations.size());
      o.first_iteration_index = n_it;
    }
    for (size_t k = 0; k < L.Iterations.size(); ++k, ++n_it) {
      if (n_it >= cap_iters) continue;
      const dvo::DenseTracker::IterationStats& I = L.Iterations[k];
      oracle_iteration_stats& o = iters[n_it];
      o.id = int(I.Id);
      o.valid_constraints = int(I.ValidConstraints);
      o.tdist_loglik = I.TDistributionLogLikelihood;
      o.tdist_mean[0] = I.TDistributionMean(0);
      o.tdist_mean[1] = I.TDistributionMean(1);
      for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b) o.tdist_precision[a * 2 + b] = I.TDistributionPrecision(a, b);
      o.prior_loglik = I.PriorLogLikelihood;
      for (int a = 0; a < 6; ++a) o.increment[a] = I.EstimateIncrement(a);
      for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 6; ++b) o.information[a * 6 + b] = I.EstimateInformation(a, b);
    }
  }
  result->n_iterations_total = n_it;
  return 0;
}

// Throughput of the reference: n_matches DenseTracker::match() calls (match k aligns pair k % n_pairs) on `nthreads` host threads,
// one tracker per thread, one match per thread at a time -- the reference's own threading model
// (dvo_slam/src/keyframe_graph.cpp:576-593).  Pyramids, derivative planes and acceleration structures are built beforehand and
// shared read-only; the point selection of the reference frame is part of every match, as in the reference.  Returns the wall
// seconds of the match phase; results[k] (k < n_pairs) = last result of pair k.
double ref_match_batch(int n_pairs, int w, int h, const float K[4], const float* const* ref_intensity, const float* const* ref_depth,
                       const float* const* cur_intensity, const float* const* cur_depth, const oracle_config* cfg, oracle_result* results,
                       int n_matches, int nthreads) {
  IntrinsicMatrix intrinsics = IntrinsicMatrix::create(K[0], K[1], K[2], K[3]);
  RgbdCameraPyramid camera(size_t(w), size_t(h), intrinsics);
  dvo::DenseTracker::Config c = dvo::DenseTracker::getDefaultConfig();
  c.FirstLevel = cfg->first_level;
  c.LastLevel = cfg->last_level;
  c.MaxIterationsPerLevel = cfg->max_iterations_per_level;
  c.UseInitialEstimate = cfg->use_initial_estimate != 0;
  c.Precision = cfg->precision;
  c.Mu = cfg->mu;
  c.IntensityDerivativeThreshold = cfg->intensity_derivative_threshold;
  c.DepthDerivativeThreshold = cfg->depth_derivative_threshold;
  camera.build(c.getNumLevels());
  auto pyramid = [&](const float* I, const float* Z) {
    cv::Mat mi(h, w, CV_32FC1), mz(h, w, CV_32FC1);
    std::memcpy(mi.data, I, size_t(w) * h * sizeof(float));
    std::memcpy(mz.data, Z, size_t(w) * h * sizeof(float));
    RgbdImagePyramidPtr p = camera.create(mi, mz);
    p->build(c.getNumLevels());
    for (size_t l = 0; l < c.getNumLevels(); ++l) {
      p->level(l).buildPointCloud();
      p->level(l).buildAccelerationStructure();
    }
    return p;
  };
  std::vector<RgbdImagePyramidPtr> refs, curs;
  for (int i = 0; i < n_pairs; ++i) {
    refs.push_back(pyramid(ref_intensity[i], ref_depth[i]));
    curs.push_back(pyramid(cur_intensity[i], cur_depth[i]));
  }
  std::vector<dvo::DenseTracker::Result> init(static_cast<size_t>(n_pairs)), last(static_cast<size_t>(n_pairs));
  for (int p = 0; p < n_pairs; ++p)
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) init[size_t(p)].Transformation.matrix()(i, j) = results[p].transformation[i * 4 + j];
  if (nthreads < 1) nthreads = 1;
  std::atomic<int> next(0);
  auto work = [&]() {
    dvo::DenseTracker tracker(c);
    for (int k = next.fetch_add(1); k < n_matches; k = next.fetch_add(1)) {
      const size_t p = size_t(k % n_pairs);
      dvo::DenseTracker::Result r;
      r.Transformation = init[p].Transformation;
      tracker.match(*refs[p], *curs[p], r);
      if (k >= n_matches - n_pairs) last[p] = r;          // exactly one writer per pair
    }
  };
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> threads;
  for (int t = 1; t < nthreads; ++t) threads.emplace_back(work);
  work();
  for (auto& t : threads) t.join();
  const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  for (int p = 0; p < n_pairs && p < n_matches; ++p) {
    const dvo::DenseTracker::Result& r = last[size_t(p)];
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) results[p].transformation[i * 4 + j] = r.Transformation.matrix()(i, j);
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) results[p].information[i * 6 + j] = r.Information(i, j);
    results[p].loglik = r.LogLikelihood;
    results[p].n_levels = int(r.Statistics.Levels.size());
    int n_it = 0;
    for (size_t l = 0; l < r.Statistics.Levels.size(); ++l) n_it += int(r.Statistics.Levels[l].Iterations.size());
    results[p].n_iterations_total = n_it;
  }
  return seconds;
}

// The reference's ConstraintProposalValidator (dvo_slam/src/constraints/*.cpp) set up as KeyframeGraph does it
// (dvo_slam/src/keyframe_graph.cpp:500-522, 819-838) on n_kf keyframes given as float planes: stage 1 = level first_level only
// with the five voters, stage 2 = first_level -> 1 with three, thresholds passed in.  Every keyframe's evaluation baseline is the
// log-likelihood of its own odometry to the neighbouring keyframe (keyframe_tracker.cpp:86-96).  Proposals: the LAST keyframe
// against every other, with identity and with the relative map pose as initial guess (keyframe_graph.cpp:576-588).
// out: per surviving proposal {reference id, current id, total score, T[16]} = 19 doubles.  Returns their number.
int ref_validate(int n_kf, int w, int h, const float K[4], const float* const* intensity, const float* const* depth, const double* poses /* n x 16 */,
                 const oracle_config* odometry_cfg, double min_constraint_ratio, double min_entropy_coarse, double min_entropy_fine,
                 double cross_validation_threshold, double* out, int cap) {
  using namespace dvo_slam;
  using namespace dvo_slam::constraints;
  auto config = [&](int first, int last) {
    dvo::DenseTracker::Config c = dvo::DenseTracker::getDefaultConfig();
    c.FirstLevel = first;
    c.LastLevel = last;
    c.Precision = odometry_cfg->precision;
    c.UseInitialEstimate = true;
    c.Mu = odometry_cfg->mu;
    c.IntensityDerivativeThreshold = odometry_cfg->intensity_derivative_threshold;
    c.DepthDerivativeThreshold = odometry_cfg->depth_derivative_threshold;
    return c;
  };
  dvo::DenseTracker::Config odometry = config(odometry_cfg->first_level, odometry_cfg->last_level);
  odometry.MaxIterationsPerLevel = odometry_cfg->max_iterations_per_level;
  IntrinsicMatrix intrinsics = IntrinsicMatrix::create(K[0], K[1], K[2], K[3]);
  RgbdCameraPyramid camera(size_t(w), size_t(h), intrinsics);
  camera.build(odometry.getNumLevels());
  std::vector<KeyframePtr> keyframes;
  for (int k = 0; k < n_kf; ++k) {
    cv::Mat mi(h, w, CV_32FC1), mz(h, w, CV_32FC1);
    std::memcpy(mi.data, intensity[k], size_t(w) * h * sizeof(float));
    std::memcpy(mz.data, depth[k], size_t(w) * h * sizeof(float));
    KeyframePtr kf(new Keyframe());
    Eigen::Affine3d pose;
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) pose.matrix()(i, j) = poses[size_t(k) * 16 + i * 4 + j];
    kf->id(short(k)).image(camera.create(mi, mz)).pose(pose);
    keyframes.push_back(kf);
  }
  dvo::DenseTracker tracker(odometry);
  for (int k = 0; k < n_kf; ++k) {
    const int other = k + 1 < n_kf ? k + 1 : k - 1;
    dvo::DenseTracker::Result r;
    r.setIdentity();
    tracker.match(*keyframes[size_t(k)]->image(), *keyframes[size_t(other)]->image(), r);
    keyframes[size_t(k)]->evaluation(TrackingResultEvaluation::ConstPtr(new LogLikelihoodTrackingResultEvaluation(r)));
  }
  ConstraintProposalValidator validator;
  validator.createStage(1)
      .trackingConfig(config(odometry_cfg->first_level, odometry_cfg->first_level))
      .keepAll()
      .addVoter(new OdometryConstraintVoter())
      .addVoter(new NaNResultVoter())
      .addVoter(new ConstraintRatioVoter(min_constraint_ratio))
      .addVoter(new TrackingResultEvaluationVoter(min_entropy_coarse))
      .addVoter(new CrossValidationVoter(cross_validation_threshold));
  validator.createStage(2)
      .trackingConfig(config(odometry_cfg->first_level, 1))
      .keepBest()
      .addVoter(new NaNResultVoter())
      .addVoter(new ConstraintRatioVoter(min_constraint_ratio))
      .addVoter(new TrackingResultEvaluationVoter(min_entropy_fine));
  ConstraintProposalVector proposals;
  const KeyframePtr& newest = keyframes.back();
  for (int k = 0; k + 1 < n_kf; ++k) {
    proposals.push_back(ConstraintProposal::createWithIdentity(newest, keyframes[size_t(k)]));
    proposals.push_back(ConstraintProposal::createWithRelative(newest, keyframes[size_t(k)]));
  }
  validator.validate(proposals);
  int n_out = 0;
  for (size_t i = 0; i < proposals.size() && n_out < cap; ++i, ++n_out) {
    double* o = out + size_t(n_out) * 19;
    o[0] = proposals[i]->Reference->id();
    o[1] = proposals[i]->Current->id();
    o[2] = proposals[i]->TotalScore();
    for (int a = 0; a < 4; ++a)
      for (int b = 0; b < 4; ++b) o[3 + a * 4 + b] = proposals[i]->TrackingResult.Transformation.matrix()(a, b);
  }
  return n_out;
}

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

// The reference's tracking front end, run as dvo_slam/src/camera_keyframe_tracking.cpp / benchmark_slam.cpp:327-383 run it:
// KeyframeTracker::update() per frame (keyframe_tracker.cpp:225-245 -> LocalTracker, local_tracker.cpp:133-216 -> LocalMap,
// local_map.cpp) with the five accept criteria of keyframe_tracker.cpp:60-72.  The pose-graph back end it hands completed local
// maps to is a counting sink (ref_graph_stub.cpp).  out_pose: n x 16, out_maps: completed local maps after each frame.
extern int g_ref_completed_local_maps;
int ref_frontend(int n, int w, int h, const float K[4], const float* const* intensity, const float* const* depth, const oracle_config* tracking,
                 double max_translational_distance, double min_entropy_ratio, double min_constraint_ratio, double* out_pose, int* out_maps) {
  dvo::DenseTracker::Config cfg = dvo::DenseTracker::getDefaultConfig();
  cfg.FirstLevel = tracking->first_level;
  cfg.LastLevel = tracking->last_level;
  cfg.MaxIterationsPerLevel = tracking->max_iterations_per_level;
  cfg.Precision = tracking->precision;
  cfg.UseInitialEstimate = tracking->use_initial_estimate != 0;
  cfg.Mu = tracking->mu;
  cfg.IntensityDerivativeThreshold = tracking->intensity_derivative_threshold;
  cfg.DepthDerivativeThreshold = tracking->depth_derivative_threshold;
  dvo_slam::KeyframeTrackerConfig selection;
  selection.MaxTranslationalDistance = max_translational_distance;
  selection.MinEntropyRatio = min_entropy_ratio;
  selection.MinEquationSystemConstraintRatio = min_constraint_ratio;
  IntrinsicMatrix intrinsics = IntrinsicMatrix::create(K[0], K[1], K[2], K[3]);
  RgbdCameraPyramid camera(size_t(w), size_t(h), intrinsics);
  camera.build(cfg.getNumLevels());
  dvo_slam::KeyframeTracker tracker;
  tracker.configureTracking(cfg);
  tracker.configureKeyframeSelection(selection);
  tracker.init();
  g_ref_completed_local_maps = 0;
  for (int k = 0; k < n; ++k) {
    cv::Mat mi(h, w, CV_32FC1), mz(h, w, CV_32FC1);
    std::memcpy(mi.data, intensity[k], size_t(w) * h * sizeof(float));
    std::memcpy(mz.data, depth[k], size_t(w) * h * sizeof(float));
    Eigen::Affine3d pose;
    pose.setIdentity();
    tracker.update(camera.create(mi, mz), ros::Time(double(k + 1) / 30.0), pose);
    for (int a = 0; a < 4; ++a)
      for (int b = 0; b < 4; ++b) out_pose[size_t(k) * 16 + a * 4 + b] = pose.matrix()(a, b);
    out_maps[k] = g_ref_completed_local_maps;
  }
  return 0;
}

}  // extern "C"
