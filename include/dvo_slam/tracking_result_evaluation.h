// dvo_slam/tracking_result_evaluation.h -- scalar quality measures of a DenseTracker::Result relative to the first /
// average result seen for a keyframe: entropy (log det Information), negative log-likelihood, and the latter per
// constraint.  The keyframe front-end polls these every frame and the loop-closure voters use them as acceptance
// ratios.  Same interface as the reference (dvo_slam/include/dvo_slam/tracking_result_evaluation.h:31-80,
// dvo_slam/src/tracking_result_evaluation.cpp:27-62); the Information matrix is a by-product of the device reduce.
#pragma once

#include <cmath>
#include <memory>

#include "dvo/dense_tracking.h"

namespace dvo_slam {

// Base: keeps the first value and the running mean of the values added since; ratios of a new result against either.
class TrackingResultEvaluation {
 public:
  using Result = dvo::DenseTracker::Result;
  typedef std::shared_ptr<TrackingResultEvaluation> Ptr;
  typedef std::shared_ptr<const TrackingResultEvaluation> ConstPtr;
  virtual ~TrackingResultEvaluation() {}

  virtual void add(const Result& r) {
    sum_ += value(r);
    count_ += 1.0;
  }
  virtual double ratioWithFirst(const Result& r) const { return value(r) / first_; }
  virtual double ratioWithAverage(const Result& r) const { return value(r) / sum_ * count_; }

 protected:
  explicit TrackingResultEvaluation(double first) : first_(first), sum_(first), count_(1.0) {}
  virtual double value(const Result& r) const = 0;

 private:
  double first_, sum_, count_;
};

namespace detail {

// one concrete evaluation per scalar measure of a result
template <typename Measure>
class MeasuredEvaluation : public TrackingResultEvaluation {
 public:
  explicit MeasuredEvaluation(const Result& first) : TrackingResultEvaluation(Measure::of(first)) {}
  double value(const Result& r) const override { return Measure::of(r); }
};

struct NegativeLogLikelihood {          // tracking_result_evaluation.cpp:54-57
  static double of(const dvo::DenseTracker::Result& r) { return -r.LogLikelihood; }
};
struct NegativeLogLikelihoodPerConstraint {   // :59-62
  static double of(const dvo::DenseTracker::Result& r) {
    return -r.LogLikelihood / double(r.Statistics.Levels.back().Iterations.back().ValidConstraints);
  }
};
struct InformationEntropy {             // :49-52
  static double of(const dvo::DenseTracker::Result& r) { return std::log(dvo::compat::determinant6(r.Information)); }
};

}  // namespace detail

// the reference's three evaluations, by name
typedef detail::MeasuredEvaluation<detail::NegativeLogLikelihood> LogLikelihoodTrackingResultEvaluation;
typedef detail::MeasuredEvaluation<detail::NegativeLogLikelihoodPerConstraint> NormalizedLogLikelihoodTrackingResultEvaluation;
typedef detail::MeasuredEvaluation<detail::InformationEntropy> EntropyRatioTrackingResultEvaluation;

}  // namespace dvo_slam
