// dvo/core/point_selection.h -- PointSelection and its predicates (dvo_core/include/dvo/core/point_selection.h:40-127).
// The compacted point list of the reference (48 B per selected point) never exists here: the predicate is folded
// into the reference-side device plane, so select() returns only the count.
#pragma once

#include "rgbd_image.h"

namespace dvo {
namespace core {

class PointSelectionPredicate {
 public:
  virtual ~PointSelectionPredicate() {}
  virtual float intensityThreshold() const { return 0.0f; }
  virtual float depthThreshold() const { return 0.0f; }
};

class ValidPointPredicate : public PointSelectionPredicate {};

class ValidPointAndGradientThresholdPredicate : public PointSelectionPredicate {
 public:
  float intensity_threshold;
  float depth_threshold;
  ValidPointAndGradientThresholdPredicate() : intensity_threshold(0.0f), depth_threshold(0.0f) {}
  virtual float intensityThreshold() const { return intensity_threshold; }
  virtual float depthThreshold() const { return depth_threshold; }
};

class PointSelection {
 public:
  explicit PointSelection(const PointSelectionPredicate& predicate) : pyramid_(0), predicate_(predicate) {}
  PointSelection(RgbdImagePyramid& pyramid, const PointSelectionPredicate& predicate) : pyramid_(&pyramid), predicate_(predicate) {}
  RgbdImagePyramid& getRgbdImagePyramid() {
    assert(pyramid_ != 0);
    return *pyramid_;
  }
  void setRgbdImagePyramid(RgbdImagePyramid& pyramid) { pyramid_ = &pyramid; }   // the device cache is keyed by (frame, level, thresholds)
  void recycle(RgbdImagePyramid& pyramid) { setRgbdImagePyramid(pyramid); }
  size_t getMaximumNumberOfPoints(const size_t& level) {   // point_selection.cpp:68-71
    const RgbdCamera& c = pyramid_->cameraPyramid().level(0);
    double n = double(c.width() * c.height());
    for (size_t l = 0; l < level; ++l) n *= 0.25;
    return size_t(n);
  }
  // number of selected reference pixels at `level` (point_selection.cpp:89-117)
  size_t select(const size_t& level) {
    assert(pyramid_ != 0);
    pyramid_->compute(level + 1);
    int n = 0;
    dvo_hip_check(pyramid_->device_context(), dvo_hip_frame_select(pyramid_->device_context(), pyramid_->device_frame(), int(level),
                  predicate_.intensityThreshold(), predicate_.depthThreshold(), &n, 0), "dvo_hip_frame_select");
    return size_t(n);
  }
  const PointSelectionPredicate& predicate() const { return predicate_; }

 private:
  RgbdImagePyramid* pyramid_;
  const PointSelectionPredicate& predicate_;
};

}  // namespace core
}  // namespace dvo
