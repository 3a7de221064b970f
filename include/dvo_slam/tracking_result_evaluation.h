// dvo_slam/tracking_result_evaluation.h -- scalar quality measures of a DenseTracker::Result relative to the first /
// average result seen for a keyframe: entropy (log det Information), negative log-likelihood, and the latter per
// constraint.  The keyframe front-end polls these every frame and the loop-closure voters use them as acceptance
// ratios.  Same interface as the reference (dvo_slam/include/dvo_slam/tracking_result_evaluation.h:31-83,
// dvo_slam/src/tracking_result_evaluation.cpp:27-62); the Information matrix is a by-product of the device reduce.
#pragma once

#include <cmath>
#include <memory>

#include "dvo/dense_tracking.h"

namespace dvo_slam {

class TrackingResultEvaluation {
 public:
  typedef std::shared_ptr<TrackingResultEvaluation> Ptr;
  typedef std::shared_ptr<const TrackingResultEvaluation> ConstPtr;
  virtual ~TrackingResultEvaluation() {}

  virtual void add(const dvo::DenseTracker::Result& r) {
    sum_ += value(r);
    count_ += 1.0;
  }
  virtual double ratioWithFirst(const dvo::DenseTracker::Result& r) const { return value(r) / first_; }
  virtual double ratioWithAverage(const dvo::DenseTracker::Result& r) const { return value(r) / sum_ * count_; }

 protected:
  explicit TrackingResultEvaluation(double first) : first_(first), sum_(first), count_(1.0) {}
  virtual double value(const dvo::DenseTracker::Result& r) const = 0;

 private:
  double first_, sum_, count_;
};

class LogLikelihoodTrackingResultEvaluation : public TrackingResultEvaluation {
 public:
  explicit LogLikelihoodTrackingResultEvaluation(const dvo::DenseTracker::Result& r) : TrackingResultEvaluation(measure(r)) {}
  virtual double value(const dvo::DenseTracker::Result& r) const { return measure(r); }

 private:
  static double measure(const dvo::DenseTracker::Result& r) { return -r.LogLikelihood; }
};

class NormalizedLogLikelihoodTrackingResultEvaluation : public TrackingResultEvaluation {
 public:
  explicit NormalizedLogLikelihoodTrackingResultEvaluation(const dvo::DenseTracker::Result& r) : TrackingResultEvaluation(measure(r)) {}
  virtual double value(const dvo::DenseTracker::Result& r) const { return measure(r); }

 private:
  static double measure(const dvo::DenseTracker::Result& r) {
    return -r.LogLikelihood / double(r.Statistics.Levels.back().Iterations.back().ValidConstraints);
  }
};

class EntropyRatioTrackingResultEvaluation : public TrackingResultEvaluation {
 public:
  explicit EntropyRatioTrackingResultEvaluation(const dvo::DenseTracker::Result& r) : TrackingResultEvaluation(measure(r)) {}
  virtual double value(const dvo::DenseTracker::Result& r) const { return measure(r); }

 private:
  static double measure(const dvo::DenseTracker::Result& r) { return std::log(dvo::compat::determinant6(r.Information)); }
};

}  // namespace dvo_slam
