// dvo/dense_tracking.h -- dvo::DenseTracker with the public surface of dvo_core/include/dvo/dense_tracking.h:36-215,
// implemented on libdvo_hip (include/dvo_hip.h).  Source-compatible with the reference's callers:
//   dvo_benchmark/src/benchmark_slam.cpp:392-415, 486      dvo_slam/src/local_tracker.cpp:59-74, 111-216
//   dvo_slam/src/constraints/constraint_proposal_validator.cpp:134-146   dvo_ros/src/camera_dense_tracking.cpp:243-269
// Semantics kept: match() always returns true (failure = Result::isNaN() / TerminationCriteria), the transformation
// argument is in/out (initial guess when UseInitialEstimate), statistics are appended, a tracker is not re-entrant.
#pragma once

#include <cassert>
#include <vector>

#include "../dvo_hip.h"
#include "core/datatypes.h"
#include "core/point_selection.h"
#include "core/rgbd_image.h"

namespace dvo {

// enums kept only so that Config compiles where the reference's callers set them; match() never reads them (Q14)
namespace core {
struct InfluenceFunctions { enum enum_t { Tukey, TDistribution, Huber, Unit }; };
struct ScaleEstimators { enum enum_t { Unit, TDistribution, MAD, NormalDistribution }; };
}  // namespace core

class DenseTracker {
 public:
  struct Config {
    int FirstLevel, LastLevel;
    int MaxIterationsPerLevel;
    double Precision;
    double Mu;
    bool UseInitialEstimate;
    bool UseWeighting;
    bool UseParallel;
    core::InfluenceFunctions::enum_t InfluenceFuntionType;
    float InfluenceFunctionParam;
    core::ScaleEstimators::enum_t ScaleEstimatorType;
    float ScaleEstimatorParam;
    float IntensityDerivativeThreshold;
    float DepthDerivativeThreshold;
    // dense_tracking_config.cpp:27-42
    Config() : FirstLevel(3), LastLevel(1), MaxIterationsPerLevel(100), Precision(5e-7), Mu(0), UseInitialEstimate(false),
               UseWeighting(true), UseParallel(false), InfluenceFuntionType(core::InfluenceFunctions::TDistribution),
               InfluenceFunctionParam(5.0f), ScaleEstimatorType(core::ScaleEstimators::TDistribution), ScaleEstimatorParam(5.0f),
               IntensityDerivativeThreshold(0.0f), DepthDerivativeThreshold(0.0f) {}
    size_t getNumLevels() const { return size_t(FirstLevel + 1); }
    bool UseEstimateSmoothing() const { return Mu > 1e-6; }
    bool IsSane() const { return FirstLevel >= LastLevel; }
  };

  struct TerminationCriteria {
    enum Enum { IterationsExceeded, IncrementTooSmall, LogLikelihoodDecreased, TooFewConstraints, NumCriteria };
  };

  struct IterationStats {
    size_t Id, ValidConstraints;
    double TDistributionLogLikelihood;
    dvo::compat::Vector2d TDistributionMean;
    dvo::compat::Matrix2d TDistributionPrecision;
    double PriorLogLikelihood;
    core::Vector6d EstimateIncrement;
    core::Matrix6d EstimateInformation;
  };
  typedef std::vector<IterationStats> IterationStatsVector;

  struct LevelStats {
    size_t Id, MaxValidPixels, ValidPixels;
    TerminationCriteria::Enum TerminationCriterion;
    IterationStatsVector Iterations;
    bool HasIterationWithIncrement() const {   // dense_tracking_config.cpp:138-143
      size_t min = (TerminationCriterion == TerminationCriteria::LogLikelihoodDecreased || TerminationCriterion == TerminationCriteria::TooFewConstraints) ? 2 : 1;
      return Iterations.size() >= min;
    }
    IterationStats& LastIterationWithIncrement() {
      assert(HasIterationWithIncrement());
      return TerminationCriterion == TerminationCriteria::LogLikelihoodDecreased ? Iterations[Iterations.size() - 2] : Iterations[Iterations.size() - 1];
    }
    const IterationStats& LastIterationWithIncrement() const {
      assert(HasIterationWithIncrement());
      return TerminationCriterion == TerminationCriteria::LogLikelihoodDecreased ? Iterations[Iterations.size() - 2] : Iterations[Iterations.size() - 1];
    }
    IterationStats& LastIteration() { return Iterations.back(); }
    const IterationStats& LastIteration() const { return Iterations.back(); }
  };
  typedef std::vector<LevelStats> LevelStatsVector;

  struct Stats { LevelStatsVector Levels; };

  // Extension over the reference: the statistics its keyframe front-end derives from a Result on the host for every
  // frame (dvo_slam/src/keyframe_tracker.cpp:165-196, tracking_result_evaluation.cpp:52-55,
  // constraints/constraint_proposal_voter.cpp:136-140), computed on the device with the result.
  struct KeyframeStatistics {
    double Entropy;                   // log det Information
    double ConditionNumber;           // |lambda_max / lambda_min| of Information
    double ConstraintRatio;           // ValidConstraints(last iteration) / ValidPixels(last level)
    double ConstraintRatioAccepted;   // the same for LastIterationWithIncrement, 0 if none
    KeyframeStatistics() : Entropy(std::numeric_limits<double>::quiet_NaN()), ConditionNumber(Entropy), ConstraintRatio(Entropy), ConstraintRatioAccepted(0.0) {}
  };

  struct Result {
    core::AffineTransformd Transformation;
    core::Matrix6d Information;
    double LogLikelihood;
    Stats Statistics;
    KeyframeStatistics Keyframe;
    Result() : LogLikelihood(std::numeric_limits<double>::max()) {   // dense_tracking_config.cpp:101-108
      double m[16];
      for (int i = 0; i < 16; ++i) m[i] = std::numeric_limits<double>::quiet_NaN();
      m[12] = m[13] = m[14] = 0.0; m[15] = 1.0;
      dvo::compat::affine_from_rowmajor(m, Transformation);
      Information.setIdentity();
    }
    bool isNaN() const {
      double m[16];
      dvo::compat::affine_to_rowmajor(Transformation, m);
      double s = 0;
      for (int i = 0; i < 16; ++i) s += m[i];
      return !std::isfinite(s) || !std::isfinite(Information.sum());
    }
    void setIdentity() {
      Transformation.setIdentity();
      Information.setIdentity();
      LogLikelihood = 0.0;
    }
    void clearStatistics() { Statistics.Levels.clear(); }
  };

  static const Config& getDefaultConfig() {
    static Config defaultConfig;
    return defaultConfig;
  }

  DenseTracker(const Config& config = getDefaultConfig()) : reference_selection_(selection_predicate_) { configure(config); }
  DenseTracker(const DenseTracker& other) : reference_selection_(selection_predicate_) { configure(other.configuration()); }   // copies the config only (dense_tracking.cpp:63-70)

  const Config& configuration() const { return cfg; }
  void configure(const Config& config) {
    assert(config.IsSane());
    cfg = config;
    selection_predicate_.intensity_threshold = cfg.IntensityDerivativeThreshold;
    selection_predicate_.depth_threshold = cfg.DepthDerivativeThreshold;
  }

  // dense_tracking.cpp:99-129 : the four overloads
  bool match(core::RgbdImagePyramid& reference, core::RgbdImagePyramid& current, core::AffineTransformd& transformation) {
    Result result;
    result.Transformation = transformation;
    bool success = match(reference, current, result);
    transformation = result.Transformation;
    return success;
  }
  bool match(core::PointSelection& reference, core::RgbdImagePyramid& current, core::AffineTransformd& transformation) {
    Result result;
    result.Transformation = transformation;
    bool success = match(reference, current, result);
    transformation = result.Transformation;
    return success;
  }
  bool match(core::RgbdImagePyramid& reference, core::RgbdImagePyramid& current, Result& result) {
    reference.compute(cfg.getNumLevels());
    reference_selection_.setRgbdImagePyramid(reference);
    return match(reference_selection_, current, result);
  }
  bool match(core::PointSelection& reference, core::RgbdImagePyramid& current, Result& result) {
    core::RgbdImagePyramid& ref = reference.getRgbdImagePyramid();
    ref.compute(cfg.getNumLevels());
    current.compute(cfg.getNumLevels());
    if (cfg.UseInitialEstimate) {
      assert(!result.isNaN() && "Provided initialization is NaN!");
    } else {
      result.setIdentity();
    }
    const dvo_hip_config c = c_config(reference.predicate().intensityThreshold(), reference.predicate().depthThreshold());
    const int nl = cfg.FirstLevel - cfg.LastLevel + 1;
    const int cap = nl * cfg.MaxIterationsPerLevel;
    levels_.resize(size_t(nl));
    iters_.resize(size_t(cap));
    dvo_hip_result r;
    dvo::compat::affine_to_rowmajor(result.Transformation, r.transformation);
    dvo_hip_context* ctx = current.device_context();
    core::dvo_hip_check(ctx, dvo_hip_match(ctx, ref.device_frame(), current.device_frame(), &c, &r, levels_.data(), nl, iters_.data(), cap), "dvo_hip_match");
    unpack(r, levels_.data(), iters_.data(), result);
    return true;   // dense_tracking.cpp:135, 375
  }


  // ---- extension over the reference API ---------------------------------------------------------------------------
  // N independent alignments in ONE device batch: what the reference runs as N match() calls on a TBB pool
  // (dvo_slam/src/keyframe_graph.cpp:576-593) or in a loop (constraint_proposal_validator.cpp:134-146).  Per pair the
  // semantics are exactly those of match(reference, current, result): results[i].Transformation is in/out, statistics
  // are appended.  Frames may repeat, in either role.
  bool matchBatch(const std::vector<core::RgbdImagePyramid*>& references, const std::vector<core::RgbdImagePyramid*>& currents,
                  const std::vector<Result*>& results) {
    assert(references.size() == currents.size() && references.size() == results.size());
    const size_t n = references.size();
    if (n == 0) return true;
    std::vector<dvo_hip_frame*> refs(n), curs(n);
    std::vector<dvo_hip_result> out(n);
    for (size_t i = 0; i < n; ++i) {
      references[i]->compute(cfg.getNumLevels());
      currents[i]->compute(cfg.getNumLevels());
      refs[i] = references[i]->device_frame();
      curs[i] = currents[i]->device_frame();
      if (cfg.UseInitialEstimate) {
        assert(!results[i]->isNaN() && "Provided initialization is NaN!");
      } else {
        results[i]->setIdentity();
      }
      dvo::compat::affine_to_rowmajor(results[i]->Transformation, out[i].transformation);
    }
    const dvo_hip_config c = c_config(selection_predicate_.intensity_threshold, selection_predicate_.depth_threshold);
    const int nl = cfg.FirstLevel - cfg.LastLevel + 1;
    const int cap = nl * cfg.MaxIterationsPerLevel;
    levels_.resize(n * size_t(nl));
    iters_.resize(n * size_t(cap));
    dvo_hip_context* ctx = currents[0]->device_context();
    core::dvo_hip_check(ctx, dvo_hip_match_batch(ctx, int(n), refs.data(), curs.data(), &c, out.data(), levels_.data(), nl, iters_.data(), cap),
                        "dvo_hip_match_batch");
    for (size_t i = 0; i < n; ++i) unpack(out[i], &levels_[i * size_t(nl)], &iters_[i * size_t(cap)], *results[i]);
    return true;
  }

 private:
  dvo_hip_config c_config(float intensity_threshold, float depth_threshold) const {
    dvo_hip_config c;
    c.first_level = cfg.FirstLevel;
    c.last_level = cfg.LastLevel;
    c.max_iterations_per_level = cfg.MaxIterationsPerLevel;
    c.use_initial_estimate = cfg.UseInitialEstimate ? 1 : 0;
    c.precision = cfg.Precision;
    c.mu = cfg.Mu;
    c.intensity_derivative_threshold = intensity_threshold;
    c.depth_derivative_threshold = depth_threshold;
    return c;
  }

  // C-ABI result + flat statistics -> Result (statistics appended, not cleared: Q15)
  static void unpack(const dvo_hip_result& r, const dvo_hip_level_stats* levels, const dvo_hip_iteration_stats* iters, Result& result) {
    dvo::compat::affine_from_rowmajor(r.transformation, result.Transformation);
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) result.Information(i, j) = r.information[i * 6 + j];
    result.LogLikelihood = r.loglik;
    result.Keyframe.Entropy = r.entropy;
    result.Keyframe.ConditionNumber = r.condition_number;
    result.Keyframe.ConstraintRatio = r.constraint_ratio;
    result.Keyframe.ConstraintRatioAccepted = r.constraint_ratio_accepted;
    for (int l = 0; l < r.n_levels; ++l) {
      result.Statistics.Levels.push_back(LevelStats());
      LevelStats& ls = result.Statistics.Levels.back();
      const dvo_hip_level_stats& s = levels[l];
      ls.Id = size_t(s.id); ls.MaxValidPixels = size_t(s.max_valid_pixels); ls.ValidPixels = size_t(s.valid_pixels);
      ls.TerminationCriterion = s.termination < 0 ? TerminationCriteria::NumCriteria : TerminationCriteria::Enum(s.termination);
      for (int k = 0; k < s.n_iterations; ++k) {
        const dvo_hip_iteration_stats& it = iters[s.first_iteration_index + k];
        IterationStats o;
        o.Id = size_t(it.id); o.ValidConstraints = size_t(it.valid_constraints);
        o.TDistributionLogLikelihood = it.tdist_loglik;
        o.TDistributionMean(0) = it.tdist_mean[0]; o.TDistributionMean(1) = it.tdist_mean[1];
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) o.TDistributionPrecision(a, b) = it.tdist_precision[a * 2 + b];
        o.PriorLogLikelihood = it.prior_loglik;
        for (int a = 0; a < 6; ++a) o.EstimateIncrement(a) = it.increment[a];
        for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) o.EstimateInformation(a, b) = it.information[a * 6 + b];
        ls.Iterations.push_back(o);
      }
    }
  }

  Config cfg;
  core::ValidPointAndGradientThresholdPredicate selection_predicate_;
  core::PointSelection reference_selection_;
  std::vector<dvo_hip_level_stats> levels_;
  std::vector<dvo_hip_iteration_stats> iters_;
};

}  // namespace dvo
