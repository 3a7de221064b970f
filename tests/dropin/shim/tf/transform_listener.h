// tests/dropin/shim (see boost/thread/mutex.hpp): TEST INFRASTRUCTURE -- the sliver of tf that dvo_ros/src/camera_dense_tracking.cpp
// and camera_base.cpp (the reference's live-camera front end, compiled unmodified against this engine's facade) need to compile and to
// be driven by a test without a ROS master.  Not ROS.
#pragma once
#include <string>
#include <ros/ros.h>
#include <tf/tf.h>
namespace tf {
class TransformListener {   // no transforms are ever available: callers fall back to the identity (dvo_ros/util/util.h:33-49)
 public:
  bool waitForTransform(const std::string&, const std::string&, const ros::Time&, const ros::Duration&, const ros::Duration&) { return false; }
  void lookupTransform(const std::string&, const std::string&, const ros::Time&, StampedTransform&) {}
};
}  // namespace tf
