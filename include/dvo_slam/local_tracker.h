// dvo_slam/local_tracker.h -- the tracking front-end: every new frame is aligned against the current KEYFRAME and
// against the PREVIOUS frame; callbacks decide whether the keyframe is still good or a new local map starts.
//
// Interface and control flow of the reference's LocalTracker (dvo_slam/include/dvo_slam/local_tracker.h:41-106,
// src/local_tracker.cpp:59-221).  The reference runs the two alignments on two DenseTracker instances under
// tbb::parallel_invoke (local_tracker.cpp:176-184); both read the same current frame.  Here they are ONE device batch of
// two pairs (DenseTracker::matchBatch): the current frame's pyramid and sampling planes are built once and swept by both.
// boost::signals2 is replaced by plain callback lists with the reference's "all slots must agree" combiner (:46-70).
#pragma once

#include <functional>
#include <memory>
#include <vector>

#include "dvo/dense_tracking.h"
#include "dvo_slam/local_map.h"

namespace dvo_slam {

class LocalTracker {
 public:
  typedef dvo::DenseTracker::Result TrackingResult;
  typedef std::function<bool(const LocalTracker&, const TrackingResult& r_odometry, const TrackingResult& r_keyframe)> AcceptCallback;
  typedef std::function<void(const LocalTracker&, const LocalMap::Ptr&, const TrackingResult& r_odometry)> MapInitializedCallback;
  typedef std::function<void(const LocalTracker&, const LocalMap::Ptr&)> MapCompleteCallback;

  LocalTracker() : force_(false) { last_keyframe_pose_.setIdentity(); }
  virtual ~LocalTracker() {}

  LocalMap::Ptr getLocalMap() const { return local_map_; }
  void getCurrentPose(dvo::core::AffineTransformd& pose) { local_map_->getCurrentFramePose(pose); }
  const dvo::DenseTracker::Config& configuration() const { return tracker_.configuration(); }
  void configure(const dvo::DenseTracker::Config& config) { tracker_.configure(config); }

  void addAcceptCallback(const AcceptCallback& cb) { accept_.push_back(cb); }
  void addMapInitializedCallback(const MapInitializedCallback& cb) { map_initialized_.push_back(cb); }
  void addMapCompleteCallback(const MapCompleteCallback& cb) { map_complete_.push_back(cb); }

  // the next update() closes the current local map whatever the callbacks say
  void forceCompleteCurrentLocalMap() { force_ = true; }

  void initNewLocalMap(const dvo::core::RgbdImagePyramid::Ptr& keyframe, const dvo::core::RgbdImagePyramid::Ptr& frame,
                       const dvo::core::AffineTransformd& keyframe_pose = dvo::core::AffineTransformd::Identity()) {
    TrackingResult r_odometry;
    r_odometry.Transformation.setIdentity();
    tracker_.match(*keyframe, *frame, r_odometry);
    last_keyframe_pose_ = r_odometry.Transformation;
    initNewLocalMap(keyframe, frame, r_odometry, keyframe_pose);
  }

  void update(const dvo::core::RgbdImagePyramid::Ptr& image, dvo::core::AffineTransformd& pose) {
    // prepare the image (local_tracker.cpp:159-170): pyramid, then per level the sampling planes -- asynchronous on the device
    const dvo::DenseTracker::Config& config = tracker_.configuration();
    image->build(config.getNumLevels());
    for (int idx = config.LastLevel; idx <= config.FirstLevel; ++idx) {
      image->level(size_t(idx)).buildPointCloud();
      image->level(size_t(idx)).buildAccelerationStructure();
    }

    TrackingResult r_odometry, r_keyframe;
    r_odometry.Transformation.setIdentity();
    r_keyframe.Transformation = last_keyframe_pose_.inverse();

    // keyframe -> image and previous frame -> image, one batch (local_tracker.cpp:172-184)
    std::vector<dvo::core::RgbdImagePyramid*> refs, curs;
    std::vector<TrackingResult*> results;
    refs.push_back(local_map_->getKeyframe().get());     curs.push_back(image.get()); results.push_back(&r_keyframe);
    refs.push_back(local_map_->getCurrentFrame().get()); curs.push_back(image.get()); results.push_back(&r_odometry);
    tracker_.matchBatch(refs, curs, results);

    force_ = force_ || r_odometry.isNaN() || r_keyframe.isNaN();
    bool accepted = true;                                 // every callback is asked, all must agree (:46-70)
    for (size_t i = 0; i < accept_.size(); ++i) accepted = accept_[i](*this, r_odometry, r_keyframe) && accepted;

    if (accepted && !force_) {
      local_map_->addFrame(image);
      local_map_->addOdometryMeasurement(r_odometry.Transformation, r_odometry.Information);
      local_map_->addKeyframeMeasurement(r_keyframe.Transformation, r_keyframe.Information);
      last_keyframe_pose_ = r_keyframe.Transformation;
    } else {
      // the previous frame becomes the keyframe of a new local map that starts with this image
      force_ = false;
      LocalMap::Ptr old_map = local_map_;
      const dvo::core::AffineTransformd old_pose = old_map->getCurrentFramePose();
      for (size_t i = 0; i < map_complete_.size(); ++i) map_complete_[i](*this, old_map);
      initNewLocalMap(old_map->getCurrentFrame(), image, r_odometry, old_pose);
      last_keyframe_pose_ = r_odometry.Transformation;
    }
    local_map_->getCurrentFramePose(pose);
  }

 private:
  void initNewLocalMap(const dvo::core::RgbdImagePyramid::Ptr& keyframe, const dvo::core::RgbdImagePyramid::Ptr& frame, TrackingResult& r_odometry,
                       const dvo::core::AffineTransformd& keyframe_pose) {
    if (r_odometry.isNaN()) r_odometry.setIdentity();     // "NaN in Map Initialization" (:146-150)
    local_map_ = LocalMap::create(keyframe, keyframe_pose);
    local_map_->addFrame(frame);
    local_map_->addKeyframeMeasurement(r_odometry.Transformation, r_odometry.Information);
    for (size_t i = 0; i < map_initialized_.size(); ++i) map_initialized_[i](*this, local_map_, r_odometry);
  }

  dvo::DenseTracker tracker_;
  dvo::core::AffineTransformd last_keyframe_pose_;
  bool force_;
  LocalMap::Ptr local_map_;
  std::vector<AcceptCallback> accept_;
  std::vector<MapInitializedCallback> map_initialized_;
  std::vector<MapCompleteCallback> map_complete_;
};

}  // namespace dvo_slam
