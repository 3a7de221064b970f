// dvo_slam/local_map.h -- the frames tracked against one keyframe, with the relative-pose measurements between them.
//
// Interface of the reference's LocalMap (dvo_slam/include/dvo_slam/local_map.h:44-88) as the tracking front-end uses
// it (local_tracker.cpp:141-216): keyframe, active frame, pose of the active frame = keyframe pose * last keyframe
// measurement (local_map.cpp:196-200).  The reference stores the measurements as a g2o pose graph and can optimise it;
// graph optimisation is outside this engine's scope, so the measurements are kept as a plain edge
// list (`measurements()`) that a g2o-equipped back-end can load in place of getGraph().
#pragma once

#include <memory>
#include <vector>

#include "dvo/core/rgbd_image.h"
#include "dvo_slam/tracking_result_evaluation.h"

namespace dvo_slam {

class LocalMap {
 public:
  typedef std::shared_ptr<LocalMap> Ptr;
  typedef std::shared_ptr<const LocalMap> ConstPtr;

  // one relative-pose constraint: vertex 0 is the keyframe, vertex k >= 1 the k-th frame added
  struct Measurement {
    int from, to;
    dvo::core::AffineTransformd transformation;
    dvo::core::Matrix6d information;
  };

  static Ptr create(const dvo::core::RgbdImagePyramid::Ptr& keyframe, const dvo::core::AffineTransformd& keyframe_pose) {
    return Ptr(new LocalMap(keyframe, keyframe_pose));
  }

  dvo::core::RgbdImagePyramid::Ptr getKeyframe() { return keyframe_; }
  dvo::core::RgbdImagePyramid::Ptr getCurrentFrame() { return current_; }

  // moves the keyframe and drags along every frame measured against it (local_map.cpp:170-186)
  void setKeyframePose(const dvo::core::AffineTransformd& keyframe_pose) {
    keyframe_pose_ = keyframe_pose;
    for (size_t i = 0; i < measurements_.size(); ++i)
      if (measurements_[i].from == 0) poses_[size_t(measurements_[i].to)] = keyframe_pose_ * measurements_[i].transformation;
  }
  const dvo::core::AffineTransformd& getKeyframePose() const { return keyframe_pose_; }

  void getCurrentFramePose(dvo::core::AffineTransformd& current_pose) { current_pose = getCurrentFramePose(); }
  dvo::core::AffineTransformd getCurrentFramePose() { return poses_.back(); }

  void setEvaluation(TrackingResultEvaluation::ConstPtr& evaluation) { evaluation_ = evaluation; }
  TrackingResultEvaluation::ConstPtr getEvaluation() { return evaluation_; }

  // the new frame becomes the active one
  void addFrame(const dvo::core::RgbdImagePyramid::Ptr& frame) {
    current_ = frame;
    frames_.push_back(frame);
    poses_.push_back(poses_.empty() ? keyframe_pose_ : poses_.back());
  }
  // previous frame -> active frame
  void addOdometryMeasurement(const dvo::core::AffineTransformd& pose, const dvo::core::Matrix6d& information) {
    add(int(frames_.size()) - 1, int(frames_.size()), pose, information);
  }
  // keyframe -> active frame; also fixes the active frame's pose estimate
  void addKeyframeMeasurement(const dvo::core::AffineTransformd& pose, const dvo::core::Matrix6d& information) {
    add(0, int(frames_.size()), pose, information);
    poses_.back() = keyframe_pose_ * pose;
  }

  const std::vector<Measurement>& measurements() const { return measurements_; }
  const std::vector<dvo::core::RgbdImagePyramid::Ptr>& frames() const { return frames_; }

 private:
  LocalMap(const dvo::core::RgbdImagePyramid::Ptr& keyframe, const dvo::core::AffineTransformd& keyframe_pose)
      : keyframe_(keyframe), keyframe_pose_(keyframe_pose) {}

  void add(int from, int to, const dvo::core::AffineTransformd& T, const dvo::core::Matrix6d& information) {
    Measurement m;
    m.from = from; m.to = to; m.transformation = T; m.information = information;
    measurements_.push_back(m);
  }

  dvo::core::RgbdImagePyramid::Ptr keyframe_, current_;
  dvo::core::AffineTransformd keyframe_pose_;
  std::vector<dvo::core::RgbdImagePyramid::Ptr> frames_;
  std::vector<dvo::core::AffineTransformd> poses_;     // pose estimate of frame k (vertex k + 1)
  std::vector<Measurement> measurements_;
  TrackingResultEvaluation::ConstPtr evaluation_;
};

}  // namespace dvo_slam
