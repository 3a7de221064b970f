// tests/dropin/shim (see boost/thread/mutex.hpp): TEST INFRASTRUCTURE -- the sliver of message_filters that dvo_ros/src/camera_dense_tracking.cpp
// and camera_base.cpp (the reference's live-camera front end, compiled unmodified against this engine's facade) need to compile and to
// be driven by a test without a ROS master.  Not ROS.
#pragma once
namespace message_filters { namespace sync_policies {
template <typename A, typename B, typename C, typename D> struct ApproximateTime {
  typedef A M0; typedef B M1; typedef C M2; typedef D M3;
  explicit ApproximateTime(int) {}
};
} }
