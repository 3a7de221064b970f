// dvo_slam/keyframe.h -- a keyframe as the loop-closure code sees it: id, device-resident image pyramid, pose in the
// map and the tracking-quality baseline of its own odometry (reference: dvo_slam/include/dvo_slam/keyframe.h:39-58).
#pragma once

#include <memory>
#include <vector>

#include "dvo/core/rgbd_image.h"
#include "dvo_slam/tracking_result_evaluation.h"

namespace dvo_slam {

class Keyframe {
 public:
  Keyframe() : id_(-1) {}
  virtual ~Keyframe() {}

  short id() const { return id_; }
  Keyframe& id(short v) { id_ = v; return *this; }
  const dvo::core::RgbdImagePyramid::Ptr& image() const { return image_; }
  Keyframe& image(const dvo::core::RgbdImagePyramid::Ptr& v) { image_ = v; return *this; }
  const dvo::core::AffineTransformd& pose() const { return pose_; }
  Keyframe& pose(const dvo::core::AffineTransformd& v) { pose_ = v; return *this; }
  const TrackingResultEvaluation::ConstPtr& evaluation() const { return evaluation_; }
  Keyframe& evaluation(const TrackingResultEvaluation::ConstPtr& v) { evaluation_ = v; return *this; }
  double timestamp() const { return image_->timestamp(); }

 private:
  short id_;
  dvo::core::RgbdImagePyramid::Ptr image_;
  dvo::core::AffineTransformd pose_;
  TrackingResultEvaluation::ConstPtr evaluation_;
};

typedef std::shared_ptr<Keyframe> KeyframePtr;
typedef std::vector<KeyframePtr> KeyframeVector;

}  // namespace dvo_slam
