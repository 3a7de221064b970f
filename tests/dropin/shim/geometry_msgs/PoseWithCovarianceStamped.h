// tests/dropin/shim (see boost/thread/mutex.hpp): TEST INFRASTRUCTURE -- the sliver of geometry_msgs that dvo_ros/src/camera_dense_tracking.cpp
// and camera_base.cpp (the reference's live-camera front end, compiled unmodified against this engine's facade) need to compile and to
// be driven by a test without a ROS master.  Not ROS.
#pragma once
#include <array>
#include <memory>
#include <geometry_msgs/Pose.h>
#include <std_msgs/Header.h>
namespace geometry_msgs {
struct CovarianceArray : std::array<double, 36> { void assign(double v) { fill(v); } };
struct PoseWithCovariance { Pose pose; CovarianceArray covariance; };
struct PoseWithCovarianceStamped {
  std_msgs::Header header;
  PoseWithCovariance pose;
};
typedef std::shared_ptr<PoseWithCovarianceStamped> PoseWithCovarianceStampedPtr;
typedef std::shared_ptr<const PoseWithCovarianceStamped> PoseWithCovarianceStampedConstPtr;
}  // namespace geometry_msgs
