// dvo/dense_tracking.h -- dvo::DenseTracker with the public surface of dvo_core/include/dvo/dense_tracking.h:36-215,
// implemented on libdvo_hip (include/dvo_hip.h).  Source-compatible with the reference's callers:
//   dvo_benchmark/src/benchmark_slam.cpp:392-415, 486      dvo_slam/src/local_tracker.cpp:59-74, 111-216
//   dvo_slam/src/constraints/constraint_proposal_validator.cpp:134-146   dvo_ros/src/camera_dense_tracking.cpp:243-269
// Semantics kept: match() always returns true (failure = Result::isNaN() / TerminationCriteria), the transformation
// argument is in/out (initial guess when UseInitialEstimate), statistics are appended, a tracker is not re-entrant.
#pragma once

#include <algorithm>
#include <cassert>
#include <cmath>
#include <ostream>
#include <string>
#include <thread>
#include <vector>

#include "../dvo_hip.h"
#include "core/datatypes.h"
#include "core/point_selection.h"
#include "core/rgbd_image.h"

namespace dvo {

// Enumerations of the reference's legacy weighting (dvo_core/include/dvo/core/weight_calculation.h:107-119, 191-203), kept
// with the same enumerator order (dynamic reconfigure maps integers onto them, dvo_ros/include/dvo_ros/util/configtools.h:70-82)
// so that Config compiles and prints where the reference's callers set them; match() never reads them.
namespace core {
struct ScaleEstimators {
  typedef enum { Unit, NormalDistribution, TDistribution, MAD } enum_t;
  static const char* str(enum_t type) {
    switch (type) {
      case Unit: return "Unit";
      case NormalDistribution: return "NormalDistribution";
      case TDistribution: return "TDistribution";
      case MAD: return "MAD";
    }
    return "";
  }
};
struct InfluenceFunctions {
  typedef enum { Unit, Tukey, TDistribution, Huber } enum_t;
  static const char* str(enum_t type) {
    switch (type) {
      case Unit: return "Unit";
      case Tukey: return "Tukey";
      case TDistribution: return "TDistribution";
      case Huber: return "Huber";
    }
    return "";
  }
};
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
    // dense_tracking_config.cpp:122-135: eigenvalues of the (symmetric) information matrix in ascending order, and
    // |largest / smallest|.  The device returns the same ratio with every result on request (Result::Keyframe.ConditionNumber).
    void InformationEigenValues(core::Vector6d& eigenvalues) const {
      double a[36], ev[6];
      for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) a[i * 6 + j] = EstimateInformation(i, j);
      dvo::compat::sym6_eigenvalues(a, ev);
      std::sort(ev, ev + 6);
      for (int i = 0; i < 6; ++i) eigenvalues(i) = ev[i];
    }
    double InformationConditionNumber() const {
      core::Vector6d ev;
      InformationEigenValues(ev);
      return std::abs(ev(5) / ev(0));
    }
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
    if (core::dvo_hip_check(ctx, dvo_hip_match(ctx, ref.device_frame(), current.device_frame(), &c, &r, levels_.data(), nl, iters_.data(), cap), "dvo_hip_match"))
      unpack(r, levels_.data(), iters_.data(), result);
    else
      result = Result();   // a failed device call reads as a NaN result, the reference's failure signal (Q16)
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
    const dvo_hip_config c = c_config(selection_predicate_.intensity_threshold, selection_predicate_.depth_threshold);
    const int nl = cfg.FirstLevel - cfg.LastLevel + 1;
    const int cap = nl * cfg.MaxIterationsPerLevel;
    // Pairs are aligned on the device their frames live on (RgbdImagePyramid binds to DeviceContext::current() when it is
    // created).  Frames spread over several GPUs give one sub-batch per device, run concurrently on one host thread each --
    // the reference's own model for independent pairs (dvo_slam/src/keyframe_graph.cpp:576-593); no collective is needed
    // inside a process, the records are concatenated in the callers' order.
    std::vector<dvo_hip_context*> devices;
    std::vector<std::vector<size_t> > members;
    for (size_t i = 0; i < n; ++i) {
      references[i]->compute(cfg.getNumLevels());
      currents[i]->compute(cfg.getNumLevels());
      dvo_hip_context* ctx = currents[i]->device_context();
      assert(references[i]->device_context() == ctx && "the two frames of a pair must live on the same device");
      size_t g = 0;
      while (g < devices.size() && devices[g] != ctx) ++g;
      if (g == devices.size()) {
        devices.push_back(ctx);
        members.push_back(std::vector<size_t>());
      }
      members[g].push_back(i);
      if (cfg.UseInitialEstimate) {
        assert(!results[i]->isNaN() && "Provided initialization is NaN!");
      } else {
        results[i]->setIdentity();
      }
    }
    levels_.resize(n * size_t(nl));
    iters_.resize(n * size_t(cap));
    std::vector<dvo_hip_result> out(n);
    std::vector<int> ok(devices.size(), 0);
    auto run = [&](size_t g) {
      const std::vector<size_t>& idx = members[g];
      const size_t m = idx.size();
      std::vector<dvo_hip_frame*> refs(m), curs(m);
      std::vector<dvo_hip_result> r(m);
      std::vector<dvo_hip_level_stats> lv(m * size_t(nl));
      std::vector<dvo_hip_iteration_stats> it(m * size_t(cap));
      for (size_t k = 0; k < m; ++k) {
        refs[k] = references[idx[k]]->device_frame();
        curs[k] = currents[idx[k]]->device_frame();
        dvo::compat::affine_to_rowmajor(results[idx[k]]->Transformation, r[k].transformation);
      }
      ok[g] = dvo_hip_match_batch(devices[g], int(m), refs.data(), curs.data(), &c, r.data(), lv.data(), nl, it.data(), cap) == DVO_HIP_OK;
      if (!ok[g]) return;
      for (size_t k = 0; k < m; ++k) {
        out[idx[k]] = r[k];
        std::copy(lv.begin() + k * size_t(nl), lv.begin() + (k + 1) * size_t(nl), levels_.begin() + idx[k] * size_t(nl));
        std::copy(it.begin() + k * size_t(cap), it.begin() + (k + 1) * size_t(cap), iters_.begin() + idx[k] * size_t(cap));
      }
    };
    if (devices.size() == 1) {
      run(0);
    } else {
      std::vector<std::thread> threads;
      for (size_t g = 1; g < devices.size(); ++g) threads.emplace_back(run, g);
      run(0);
      for (size_t t = 0; t < threads.size(); ++t) threads[t].join();
    }
    for (size_t g = 0; g < devices.size(); ++g) {
      if (!ok[g]) core::dvo_hip_check(devices[g], DVO_HIP_ERR_HIP, "dvo_hip_match_batch");
      for (size_t k = 0; k < members[g].size(); ++k) {
        const size_t i = members[g][k];
        if (ok[g]) unpack(out[i], &levels_[i * size_t(nl)], &iters_[i * size_t(cap)], *results[i]);
        else *results[i] = Result();
      }
    }
    return true;
  }

  // dense_tracking.cpp:378-444 (debug aid): |intensity residual| of every selected reference pixel that yields a constraint
  // under `transformation` (reference -> current) at `level`, 0 elsewhere.  One sweep of the level on the device.
  dvo::compat::ImageMat computeIntensityErrorImage(core::RgbdImagePyramid& reference, core::RgbdImagePyramid& current,
                                                   const core::AffineTransformd& transformation, size_t level = 0) {
    reference.compute(level + 1);
    current.compute(level + 1);
    const core::RgbdCamera& cam = reference.cameraPyramid().level(level);
    const int w = int(cam.width()), h = int(cam.height());
    double m[16];
    dvo::compat::affine_to_rowmajor(transformation, m);
    float T34[12];
    for (int i = 0; i < 12; ++i) T34[i] = float(m[i]);
    const float P0[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    std::vector<float> residuals(size_t(w) * h * 2);
    dvo_hip_iteration_out out;
    dvo::compat::ImageMat result = dvo::compat::image_create(h, w);
    float* r = dvo::compat::image_ptr_mut(result);
    dvo_hip_context* ctx = current.device_context();
    const bool ok = core::dvo_hip_check(ctx, dvo_hip_level_iteration(ctx, reference.device_frame(), current.device_frame(), int(level),
                                                                     selection_predicate_.intensity_threshold, selection_predicate_.depth_threshold,
                                                                     T34, P0, 1, &out, residuals.data()), "dvo_hip_level_iteration");
    for (size_t i = 0; i < size_t(w) * h; ++i) {
      const float r0 = residuals[2 * i];
      r[i] = (ok && r0 == r0) ? std::fabs(r0) : 0.0f;
    }
    return result;
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

// ---- printers (dvo_core/include/dvo/dense_tracking.h:218-291; used by dvo_slam/src/keyframe_graph.cpp:360 and
// dvo_benchmark/src/benchmark_slam.cpp:413) -- same fields in the same order and wording, so logs stay comparable ------------
template <typename CharT, typename Traits>
std::basic_ostream<CharT, Traits>& operator<<(std::basic_ostream<CharT, Traits>& out, const dvo::DenseTracker::Config& config) {
  out << "First Level = " << config.FirstLevel << ", Last Level = " << config.LastLevel
      << ", Max Iterations per Level = " << config.MaxIterationsPerLevel << ", Precision = " << config.Precision
      << ", Mu = " << config.Mu << ", Use Initial Estimate = " << (config.UseInitialEstimate ? "true" : "false")
      << ", Use Weighting = " << (config.UseWeighting ? "true" : "false")
      << ", Scale Estimator = " << dvo::core::ScaleEstimators::str(config.ScaleEstimatorType)
      << ", Scale Estimator Param = " << config.ScaleEstimatorParam
      << ", Influence Function = " << dvo::core::InfluenceFunctions::str(config.InfluenceFuntionType)
      << ", Influence Function Param = " << config.InfluenceFunctionParam
      << ", Intensity Derivative Threshold = " << config.IntensityDerivativeThreshold
      << ", Depth Derivative Threshold = " << config.DepthDerivativeThreshold;
  return out;
}

template <typename CharT, typename Traits>
std::basic_ostream<CharT, Traits>& operator<<(std::basic_ostream<CharT, Traits>& o, const dvo::DenseTracker::IterationStats& s) {
  o << "Iteration: " << s.Id << " ValidConstraints: " << s.ValidConstraints << " DataLogLikelihood: " << s.TDistributionLogLikelihood
    << " PriorLogLikelihood: " << s.PriorLogLikelihood << std::endl;
  return o;
}

template <typename CharT, typename Traits>
std::basic_ostream<CharT, Traits>& operator<<(std::basic_ostream<CharT, Traits>& o, const dvo::DenseTracker::LevelStats& s) {
  static const char* const names[] = {"IterationsExceeded", "IncrementTooSmall", "LogLikelihoodDecreased", "TooFewConstraints"};
  const int t = int(s.TerminationCriterion);
  o << "Level: " << s.Id << " Pixel: " << s.ValidPixels << "/" << s.MaxValidPixels << " Termination: "
    << (t >= 0 && t < 4 ? names[t] : "") << " Iterations: " << s.Iterations.size() << std::endl;
  for (dvo::DenseTracker::IterationStatsVector::const_iterator it = s.Iterations.begin(); it != s.Iterations.end(); ++it) o << *it;
  return o;
}

template <typename CharT, typename Traits>
std::basic_ostream<CharT, Traits>& operator<<(std::basic_ostream<CharT, Traits>& o, const dvo::DenseTracker::Stats& s) {
  o << s.Levels.size() << " levels" << std::endl;
  for (dvo::DenseTracker::LevelStatsVector::const_iterator it = s.Levels.begin(); it != s.Levels.end(); ++it) o << *it;
  return o;
}
