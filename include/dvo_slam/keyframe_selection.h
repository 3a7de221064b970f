// dvo_slam/keyframe_selection.h -- when does the tracking front-end start a new keyframe?
//
// The accept criteria the reference's KeyframeTracker installs on its LocalTracker
// (dvo_slam/src/keyframe_tracker.cpp:60-72, 86-168; configuration dvo_slam/include/dvo_slam/config.h:30-41,
// src/config.cpp:27-34): tracking-quality ratio against the keyframe's first result, estimate divergence, distance to
// the keyframe, constraint ratio.  The graph back-end the reference's class also owns (g2o) is outside this engine's
// scope; this header carries the front-end half so LocalTracker can be driven the same way.  The constraint ratio is
// read from the statistics the device returns with every result (DenseTracker::Result::Keyframe).
#pragma once

#include <cmath>
#include <limits>
#include <memory>

#include "dvo_slam/local_tracker.h"
#include "dvo_slam/tracking_result_evaluation.h"

namespace dvo_slam {

struct KeyframeTrackerConfig {
  bool UseMultiThreading;
  double MaxTranslationalDistance;
  double MaxRotationalDistance;
  double MinEntropyRatio;
  double MinEquationSystemConstraintRatio;
  KeyframeTrackerConfig()
      : UseMultiThreading(true), MaxTranslationalDistance(0.2), MaxRotationalDistance(std::numeric_limits<double>::max()), MinEntropyRatio(0.91),
        MinEquationSystemConstraintRatio(0.33) {}
};

class KeyframeSelection {
 public:
  explicit KeyframeSelection(const KeyframeTrackerConfig& cfg = KeyframeTrackerConfig()) : cfg_(cfg) { last_transform_to_keyframe_.setIdentity(); }

  const KeyframeTrackerConfig& configuration() const { return cfg_; }
  void configure(const KeyframeTrackerConfig& cfg) { cfg_ = cfg; }
  // quality baseline of the current keyframe (handed to the map on completion, keyframe_tracker.cpp:98-103)
  TrackingResultEvaluation::ConstPtr evaluation() const { return evaluation_; }

  // registers the callbacks in the reference's order; `this` must outlive the tracker
  void install(LocalTracker& lt) {
    lt.addMapInitializedCallback([this](const LocalTracker&, const LocalMap::Ptr&, const LocalTracker::TrackingResult& r_odometry) {
      last_transform_to_keyframe_ = r_odometry.Transformation;
      evaluation_.reset(new LogLikelihoodTrackingResultEvaluation(r_odometry));
    });
    lt.addMapCompleteCallback([this](const LocalTracker&, const LocalMap::Ptr& m) {
      TrackingResultEvaluation::ConstPtr e(evaluation_);
      m->setEvaluation(e);
    });
    lt.addAcceptCallback([this](const LocalTracker&, const LocalTracker::TrackingResult&, const LocalTracker::TrackingResult& r_keyframe) {
      return acceptTrackingResultEvaluation(r_keyframe);
    });
    lt.addAcceptCallback([this](const LocalTracker&, const LocalTracker::TrackingResult& r_odometry, const LocalTracker::TrackingResult& r_keyframe) {
      return acceptEstimateDivergence(r_odometry, r_keyframe);
    });
    lt.addAcceptCallback([this](const LocalTracker&, const LocalTracker::TrackingResult&, const LocalTracker::TrackingResult& r_keyframe) {
      return translationNorm(r_keyframe) < cfg_.MaxTranslationalDistance;                       // keyframe_tracker.cpp:153-156
    });
    lt.addAcceptCallback([this](const LocalTracker&, const LocalTracker::TrackingResult&, const LocalTracker::TrackingResult& r_keyframe) {
      return r_keyframe.Keyframe.ConstraintRatio > cfg_.MinEquationSystemConstraintRatio;       // keyframe_tracker.cpp:165-168
    });
  }

  // keyframe_tracker.cpp:105-123: ratio of this result's quality to the keyframe's first result; accepted results
  // enter the running average
  bool acceptTrackingResultEvaluation(const LocalTracker::TrackingResult& r_keyframe) {
    const bool accept = evaluation_->ratioWithFirst(r_keyframe) > cfg_.MinEntropyRatio;
    if (accept) evaluation_->add(r_keyframe);
    return accept;
  }

  // keyframe_tracker.cpp:125-151: a jump of the odometry (> 0.1 m between consecutive frames) or of the keyframe estimate
  // is treated as a tracking failure; the reference then overwrites the (const) results in place so that the new local
  // map starts from a neutral odometry estimate and the last good keyframe transform -- reproduced as is
  bool acceptEstimateDivergence(const LocalTracker::TrackingResult& r_odometry, const LocalTracker::TrackingResult& r_keyframe) {
    const bool reject = translationNorm(r_odometry) > 0.1 || translationNorm(r_keyframe) > 1.5 * cfg_.MaxTranslationalDistance;
    if (reject) {
      LocalTracker::TrackingResult& ro = const_cast<LocalTracker::TrackingResult&>(r_odometry);
      ro.Transformation.setIdentity();
      ro.Information.setIdentity();
      for (int i = 0; i < 6; ++i) ro.Information(i, i) = 0.008 * 0.008;
      const_cast<LocalTracker::TrackingResult&>(r_keyframe).Transformation = last_transform_to_keyframe_;
    }
    last_transform_to_keyframe_ = r_keyframe.Transformation;
    return !reject;
  }

  static double translationNorm(const LocalTracker::TrackingResult& r) {
    double m[16];
    dvo::compat::affine_to_rowmajor(r.Transformation, m);
    return std::sqrt(m[3] * m[3] + m[7] * m[7] + m[11] * m[11]);
  }

 private:
  KeyframeTrackerConfig cfg_;
  TrackingResultEvaluation::Ptr evaluation_;
  dvo::core::AffineTransformd last_transform_to_keyframe_;
};

}  // namespace dvo_slam
